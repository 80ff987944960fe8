#!/bin/bash
# round 4, GPU call AJ: 4-wide layers (box heads, query_pos) through _LinearMaster: tests, module tests, bench
set -o pipefail
O=gpurun_out/r4aj; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_modules.py -q -m gpu -k "linear_master or decoder or meh_head or full_model_vs or training_step" > $O/t.txt 2>&1; echo "tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t.txt | cut -c1-300 | head -5
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench.json 2> $O/bench.err; grep -E "timed" $O/bench.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench2.json 2> $O/bench2.err; grep -E "timed" $O/bench2.err
