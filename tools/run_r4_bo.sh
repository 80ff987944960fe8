#!/bin/bash
# round 4, call BO: ops.embed_rows (denoising class-embedding lookup): tests + A/B
set -o pipefail
O=gpurun_out/r4bo; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_modules.py -x -q -m gpu -k "embed_rows or cdn or meh_head or full_model or training_step" > $O/t.txt 2>&1; echo "tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t.txt | head | cut -c1-300
for i in 1 2; do
TAMTR_EMBED_ROWS=torch timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_off$i.json 2> $O/bench_off$i.err; grep -E "timed" $O/bench_off$i.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_on$i.json 2> $O/bench_on$i.err; grep -E "timed" $O/bench_on$i.err
done
