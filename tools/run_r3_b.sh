#!/bin/bash
# round 3, GPU call B: whole GPU suite (sorted deformable backward, atomics-free scan / contrastive sums, deterministic mode, replay tests),
# default bench, bench in deterministic mode (its cost).  A step killed at its limit ends the call.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
killed() { [ "$1" = 124 ] || [ "$1" = 137 ]; }
timeout -k 10 1000 python -m pytest tests -q -m gpu -s > gpurun_out/r3b_tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee gpurun_out/r3b_status.txt; tail -15 gpurun_out/r3b_tests.log
killed $rc && exit $rc
timeout -k 10 600 python bench.py > gpurun_out/r3b_bench.json 2> gpurun_out/r3b_bench.err
rc=$?; echo "bench rc=$rc" | tee -a gpurun_out/r3b_status.txt; tail -1 gpurun_out/r3b_bench.json | cut -c1-1500
killed $rc && exit $rc
TAMTR_DETERMINISTIC=1 timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/r3b_bench_det.json 2> gpurun_out/r3b_bench_det.err
rc=$?; echo "bench(det) rc=$rc" | tee -a gpurun_out/r3b_status.txt; tail -1 gpurun_out/r3b_bench_det.json | cut -c1-1200
