#!/usr/bin/env python3
"""A/B of the value-projection GEMM kernels (TAMTR_GEMM = ws2 | ws1 | tile, read when the library loads: one process per choice):
correctness against an fp32 product of the same bf16 values on sampled rows + column checksums, then timing two ways - back to back
(sustained MFMA load: the part gives clock back) and interleaved with a memory-bound kernel (closer to its place in the step)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tamtr_amd.ops as ops
from tamtr_amd import _lib
M, N, K = int(os.environ.get('GM', 16 * 33600)), 512, 512
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn(M, K, device='cuda', generator=g).bfloat16()
w = (torch.randn(N, K, device='cuda', generator=g) * K ** -0.5).bfloat16()
b = torch.randn(N, device='cuda', generator=g)
y = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)


def run():
    _lib.call('tamtr_linear_bf16', _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), M, N, K, _lib.stream_ptr())


run()
torch.cuda.synchronize()
idx = torch.cat([torch.randint(0, M, (256,), device='cuda', generator=g), torch.tensor([0, 1, 31, 32, M // 2, M - 33, M - 32, M - 1], device='cuda')])
ref = x[idx].float() @ w.float().t() + b
err = float((y[idx].float() - ref).abs().max())
chk = float((y.float().sum(0) - ((x.float().sum(0, keepdim=True) @ w.float().t()).squeeze(0) + M * b)).abs().max()) / M ** 0.5
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for _ in range(5):
    run()
torch.cuda.synchronize()
ev[0].record()
for _ in range(50):
    run()
ev[1].record()
torch.cuda.synchronize()
b2b = ev[0].elapsed_time(ev[1]) / 50 * 1e3
big = torch.empty(1 << 28, device='cuda', dtype=torch.bfloat16)
ts = []
for _ in range(20):
    big.add_(1)            # a memory-bound neighbour
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); c.record()
    ts.append((a, c))
torch.cuda.synchronize()
iso = sorted(a.elapsed_time(c) for a, c in ts)
flops = 2.0 * M * N * K
print(json.dumps({'kernel': os.environ.get('TAMTR_GEMM', 'ws2'), 'M': M, 'sampled_rows_max_abs_err': err, 'column_checksum_err_over_sqrtM': chk,
                  'back_to_back_us': round(b2b, 1), 'interleaved_us_median': round(iso[len(iso) // 2] * 1e3, 1), 'interleaved_us_min': round(iso[0] * 1e3, 1),
                  'frac_of_2.5PF_back_to_back': round(flops / (b2b * 1e-6) / 2.5e15, 3), 'frac_interleaved': round(flops / (iso[len(iso) // 2] * 1e-3) / 2.5e15, 3)}))
