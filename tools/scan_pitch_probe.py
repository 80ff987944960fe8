#!/usr/bin/env python3
"""Does the scan suffer from L2-channel camping?  Row pitch L*4 B = 102400 B is a multiple of 4 KiB at level 0, so every row's
chunk c starts in the same L2 channels.  Time fwd / fwd+bwd at L = 25600 and at slightly different lengths."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tamtr_amd.ops as ops
from bench_kernels import timeit
B, K, N, Dk, R = 16, 4, 16, 256, 8
for L in (25600, 25664, 25728, 26624, 6400, 6464):
    g = torch.Generator(device='cuda').manual_seed(0)
    rn = lambda *s: torch.randn(*s, device='cuda', generator=g)
    ins = [t.requires_grad_() for t in (rn(B, 2, Dk, L), rn(B, K, R, L), rn(K * Dk, R) * R ** -0.5, -torch.exp(rn(K * Dk, N) * 0.3),
                                        rn(B, K, N, L), rn(B, K, N, L), rn(K * Dk), rn(K * Dk) - 3)]
    y = ops.selective_scan_cross(*ins)
    gy = torch.randn_like(y)
    f, _ = timeit(lambda: ops.selective_scan_cross(*ins), n=6)
    fb, _ = timeit(lambda: torch.autograd.grad(ops.selective_scan_cross(*ins), ins, gy), n=6)
    print(f'L={L} (row pitch {L * 4} B, mod 4096 = {L * 4 % 4096}): fwd {f:.2f} ms, bwd {fb - f:.2f} ms; per 25600 steps: fwd {f * 25600 / L:.2f}, bwd {(fb - f) * 25600 / L:.2f}')
    del ins, y, gy
    torch.cuda.empty_cache()
