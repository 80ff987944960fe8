#!/bin/bash
# round 3, GPU call C: re-run of the suites touched since call B; bench in deterministic mode with MIOpen's search among deterministic solvers
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
killed() { [ "$1" = 124 ] || [ "$1" = 137 ]; }
timeout -k 10 900 python -m pytest tests/test_gpu_graphs.py tests/test_gpu_fullsize.py tests/test_gpu_modules.py -q -m gpu -s > gpurun_out/r3c_tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee gpurun_out/r3c_status.txt; tail -12 gpurun_out/r3c_tests.log
killed $rc && exit $rc
TAMTR_DETERMINISTIC=1 timeout -k 10 900 python bench.py --no-cpu-baseline --steps 10 > gpurun_out/r3c_bench_det.json 2> gpurun_out/r3c_bench_det.err
rc=$?; echo "bench(det) rc=$rc" | tee -a gpurun_out/r3c_status.txt; tail -1 gpurun_out/r3c_bench_det.json | cut -c1-1500; tail -5 gpurun_out/r3c_bench_det.err
