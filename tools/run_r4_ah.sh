#!/bin/bash
# round 4, GPU call AH: which BLAS backend serves the step's library GEMMs best (global preference A/B)
set -o pipefail
O=gpurun_out/r4ah; mkdir -p $O
python3 -c "import torch; print('preferred:', torch.backends.cuda.preferred_blas_library())" > $O/pref.txt 2>&1; cat $O/pref.txt | tail -1
for v in default 0 1 default 0; do
  if [ "$v" = default ]; then unset TORCH_BLAS_PREFER_HIPBLASLT; else export TORCH_BLAS_PREFER_HIPBLASLT=$v; fi
  timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_$v.json 2> $O/bench_$v.err; echo "TORCH_BLAS_PREFER_HIPBLASLT=$v $(grep -E 'timed' $O/bench_$v.err | cut -c1-100)" | tee -a $O/ab.txt
done
python3 - > $O/small_gemm.txt 2>&1 <<'PY'
import torch
def t(fn, n=20):
    for _ in range(5): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[n // 2] * 1e3
for lib in ('hipblaslt', 'cublas'):
    torch.backends.cuda.preferred_blas_library(lib)
    for M, N, K in [(4672, 512, 512), (4672, 1024, 512), (4672, 512, 1024), (4672, 192, 512), (4672, 96, 512), (4672, 1536, 512)]:
        x = torch.randn(M, K, device='cuda').bfloat16(); w = torch.randn(N, K, device='cuda').bfloat16(); b = torch.randn(N, device='cuda').bfloat16(); g = torch.randn(M, N, device='cuda').bfloat16()
        f = t(lambda: torch.nn.functional.linear(x, w, b)); dx = t(lambda: torch.mm(g, w)); dw = t(lambda: torch.mm(g.t(), x))
        print(f'{lib:10s} M={M} N={N} K={K}: fwd {f:5.1f} us  dX {dx:5.1f} us  dW {dw:5.1f} us   (event bracket ~12 us included)')
PY
cat $O/small_gemm.txt
