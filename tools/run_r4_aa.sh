#!/bin/bash
# round 4, GPU call AA: sum_n with unconditional loads: test, timing, A/B by kernel time in the step
set -o pipefail
O=gpurun_out/r4aa; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "fanout or slab_sum or colsum" > $O/t.txt 2>&1; echo "tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t.txt | cut -c1-300 | head -5
python3 - > $O/sum_n.txt 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, '.')
import tamtr_amd.ops as ops
from tamtr_amd._lib import call, ptr, stream_ptr
import ctypes
x = [torch.randn(16, 33600, 256, device='cuda').bfloat16() for _ in range(4)]
out = torch.empty_like(x[0])
arr = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in x])
def go(): call('tamtr_sum_n', ctypes.cast(arr, ctypes.c_void_p), 4, ptr(out), out.numel(), 1, stream_ptr())
for _ in range(3): go()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for a, b in ev:
    a.record(); go(); b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)
byt = 5 * out.numel() * 2
print(f'sum_n 4 x [16, 33600, 256] bf16: {ms[0]*1e3:.0f} us min, {sum(ms)/len(ms)*1e3:.0f} avg = {byt/ms[0]/1e6:.0f} GB/s')
ref = (x[0].float() + x[1].float() + x[2].float() + x[3].float()).bfloat16()
print('equal to the fp32 sum rounded once:', bool(torch.equal(out, ref)))
PY
cat $O/sum_n.txt
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench.json 2> $O/bench.err; grep -E "timed" $O/bench.err
