#!/bin/bash
# round 4, call BQ: value_proj + MSDA as one node (bias gradient from colw, ABI 34): tests + A/B
set -o pipefail
O=gpurun_out/r4bq; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_modules.py tests/test_gpu_fullsize.py tests/test_cabi.py -x -q -m gpu -k "value_proj or msda or msdeform or decoder or meh_head or full_model or training_step or hip_path or abi" > $O/t.txt 2>&1; echo "tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t.txt | head | cut -c1-300
for i in 1 2; do
TAMTR_VALUE_BIAS=colsum timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_off$i.json 2> $O/bench_off$i.err; grep -E "timed" $O/bench_off$i.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_on$i.json 2> $O/bench_on$i.err; grep -E "timed" $O/bench_on$i.err
done
