#!/usr/bin/env python3
"""A few forward+backward passes of the level-0 scan (bs 16, d_inner 256, L 25600, R 8) and nothing else: the target of
rocprofv3 --pmc runs.  python3 tools/scan_only.py [level]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tamtr_amd.ops as ops
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 0
d_inner, L, R = [(256, 25600, 8), (512, 6400, 16), (1024, 1600, 32)][lvl]
B, K, N = 16, 4, 16
g = torch.Generator(device='cuda').manual_seed(0)
u2 = torch.randn(B, 2, d_inner, L, device='cuda', generator=g)
dtr = torch.randn(B, K, R, L, device='cuda', generator=g)
Wdt = torch.randn(K * d_inner, R, device='cuda', generator=g) * R ** -0.5
A = -torch.exp(torch.randn(K * d_inner, N, device='cuda', generator=g) * 0.3)
Bm = torch.randn(B, K, N, L, device='cuda', generator=g)
Cm = torch.randn(B, K, N, L, device='cuda', generator=g)
D = torch.randn(K * d_inner, device='cuda', generator=g)
bias = torch.randn(K * d_inner, device='cuda', generator=g) - 3
ins = [t.requires_grad_() for t in (u2, dtr, Wdt, A, Bm, Cm, D, bias)]
gy = torch.randn(B, K * d_inner, L, device='cuda', generator=g)
for _ in range(3):
    y = ops.selective_scan_cross(*ins)
    torch.autograd.grad(y, ins, gy)
torch.cuda.synchronize()
