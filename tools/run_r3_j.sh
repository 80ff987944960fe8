#!/bin/bash
# GPU call J: whole-token-row deformable backward (tests + microbench + A/B), deterministic bench without timed search
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
killed() { [ "$1" = 124 ] || [ "$1" = 137 ]; }
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -q -m gpu -k "msdeform or colsum" > gpurun_out/r3j_tests.log 2>&1
rc=$?; echo "msda tests rc=$rc" | tee gpurun_out/r3j_status.txt; tail -5 gpurun_out/r3j_tests.log | cut -c1-200
killed $rc && exit $rc
[ $rc != 0 ] && exit $rc
timeout -k 10 200 python tools/bench_kernels.py msda > gpurun_out/r3j_msda_rows.txt 2>&1; rc=$?; killed $rc && exit $rc
TAMTR_MSDA_PER_HEAD=1 timeout -k 10 200 python tools/bench_kernels.py msda > gpurun_out/r3j_msda_per_head.txt 2>&1; rc=$?; killed $rc && exit $rc
echo rows; tail -3 gpurun_out/r3j_msda_rows.txt; echo per-head; tail -3 gpurun_out/r3j_msda_per_head.txt
TAMTR_DETERMINISTIC=1 timeout -k 10 500 python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/r03_bench_deterministic.json 2> gpurun_out/r03_bench_deterministic.err
rc=$?; echo "bench det rc=$rc" | tee -a gpurun_out/r3j_status.txt; grep -E "timed|captured|graph vs" gpurun_out/r03_bench_deterministic.err | cut -c1-300
