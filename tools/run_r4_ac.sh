#!/bin/bash
# round 4, GPU call AC: short-row LayerNorm kernels: tests, timing at the level-0 shape, VSS block tests, bench
set -o pipefail
O=gpurun_out/r4ac; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_modules.py -q -m gpu -k "layer_norm or vss_block or ss2d_core" > $O/t.txt 2>&1; echo "tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t.txt | cut -c1-400 | head
python3 - > $O/ln_micro.txt 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, '.')
import tamtr_amd.ops as ops
def t(fn, n=10):
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return min(a.elapsed_time(b) for a, b in ev) * 1e3
for D, n in [(128, 409600), (256, 102400), (512, 25600)]:
    x = torch.randn(n, D, device='cuda').bfloat16().requires_grad_()
    g, b = torch.ones(D, device='cuda', requires_grad=True), torch.zeros(D, device='cuda', requires_grad=True)
    go = torch.randn(n, D, device='cuda').bfloat16()
    f = t(lambda: ops.layer_norm(x, g, b, 1e-5))
    fb = t(lambda: torch.autograd.grad(ops.layer_norm(x, g, b, 1e-5), [x, g, b], go))
    print(f'layer_norm bf16 [{n}, {D}]: fwd {f:.0f} us = {2*n*D*2/f/1e6:.0f} GB/s, fwd+bwd {fb:.0f} us')
PY
cat $O/ln_micro.txt | tail -3
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench.json 2> $O/bench.err; grep -E "timed" $O/bench.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench2.json 2> $O/bench2.err; grep -E "timed" $O/bench2.err
