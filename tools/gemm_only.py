import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tamtr_amd.ops as ops
M, N, K = 16 * 33600, 512, 512
x = torch.randn(M, K, device='cuda').bfloat16(); w = (torch.randn(N, K, device='cuda') * K ** -0.5); b = torch.randn(N, device='cuda')
for _ in range(6):
    ops.linear_bf16(x, w, b)
torch.cuda.synchronize()
