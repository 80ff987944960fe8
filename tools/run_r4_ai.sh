#!/bin/bash
# round 4, GPU call AI: decoder weight gradients as 8-slice batched products: test, A/B bench
set -o pipefail
O=gpurun_out/r4ai; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "linear_master" > $O/t.txt 2>&1; echo "tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t.txt | cut -c1-300 | head -3
for v in 0 512 256 0 512 256; do
  TAMTR_LINEAR_MASTER_SLICE_ROWS=$v timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_$v.json 2> $O/bench.err; echo "slice rows >= $v: $(grep -E 'timed' $O/bench.err | cut -c1-100)" | tee -a $O/ab.txt
done
