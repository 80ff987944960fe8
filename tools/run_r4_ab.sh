#!/bin/bash
# round 4, GPU call AB: channels-last CPAM kernels: tests, micro timing, A/B bench
set -o pipefail
O=gpurun_out/r4ab; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "cpam or maxpool" > $O/t.txt 2>&1; echo "tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t.txt | cut -c1-400 | head
python3 - > $O/cpam_micro.txt 2>&1 <<'PY'
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
for C, S in [(512, 40), (256, 80), (128, 160)]:
    x = torch.randn(16, C, S, S, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_()
    go = torch.randn(16, C, S, S, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
    f_cl = t(lambda: ops._CPAMCL.apply(x))
    fb_cl = t(lambda: torch.autograd.grad(ops._CPAMCL.apply(x), x, go))
    f_n = t(lambda: ops.to_channels_last(ops._CPAM.apply(ops.to_nchw(x))))
    fb_n = t(lambda: torch.autograd.grad(ops.to_channels_last(ops._CPAM.apply(ops.to_nchw(x))), x, go))
    byt = x.numel() * 2
    print(f'cpam {C} ch x {S}^2 x 16 bf16 channels-last map: own CL kernels fwd {f_cl:.0f} us, fwd+bwd {fb_cl:.0f} us; NCHW kernels behind repacks fwd {f_n:.0f} us, fwd+bwd {fb_n:.0f} us  (map = {byt/1e6:.0f} MB)')
PY
cat $O/cpam_micro.txt | tail -4
TAMTR_CPAM=nchw timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_off.json 2> $O/bench_off.err; grep -E "timed" $O/bench_off.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_on.json 2> $O/bench_on.err; grep -E "timed|graph vs" $O/bench_on.err | cut -c1-250
TAMTR_CPAM=nchw timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_off2.json 2> $O/bench_off2.err; grep -E "timed" $O/bench_off2.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_on2.json 2> $O/bench_on2.err; grep -E "timed" $O/bench_on2.err
