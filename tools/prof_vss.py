#!/usr/bin/env python3
"""One VSSBlock fwd+bwd at a MEH level, for `rocprofv3 --kernel-trace --stats -- python3 tools/prof_vss.py [level]`."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tamtr_amd.vss import VSSBlock
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 0
d, hw = [(128, 160), (256, 80), (512, 40)][lvl]
blk = VSSBlock(hidden_dim=d, drop_path=0.1).cuda().train()
x = torch.randn(16, hw, hw, d, device='cuda').bfloat16().requires_grad_()
for _ in range(5):
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y = blk(x)
    y.float().sum().backward()
torch.cuda.synchronize()
