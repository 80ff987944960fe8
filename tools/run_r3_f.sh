#!/bin/bash
# round 3, GPU call F: GEMM kernel A/B (hand-laid 64-column kernel), linear tests; deterministic tests on the NCHW trunk with durations
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
killed() { [ "$1" = 124 ] || [ "$1" = 137 ]; }
rm -f gpurun_out/r3f_gemm_ab.txt
for k in ws2 ws1; do
  TAMTR_GEMM=$k timeout -k 10 200 python3 tools/gemm_ab.py >> gpurun_out/r3f_gemm_ab.txt 2>gpurun_out/r3f_gemm_ab.err
  rc=$?; [ $rc != 0 ] && { echo "gemm_ab $k rc=$rc" | tee -a gpurun_out/r3f_status.txt; tail -5 gpurun_out/r3f_gemm_ab.err; }
  killed $rc && exit $rc
done
for m in 1075200 1000 537599; do GM=$m TAMTR_GEMM=ws2 timeout -k 10 200 python3 tools/gemm_ab.py >> gpurun_out/r3f_gemm_ab.txt 2>>gpurun_out/r3f_gemm_ab.err; done
cut -c1-420 gpurun_out/r3f_gemm_ab.txt
timeout -k 10 600 python -m pytest tests/test_gpu_modules.py -q -m gpu -k "linear" > gpurun_out/r3f_tests_linear.log 2>&1
rc=$?; echo "linear tests rc=$rc" | tee -a gpurun_out/r3f_status.txt; tail -4 gpurun_out/r3f_tests_linear.log
killed $rc && exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_graphs.py tests/test_gpu_fullsize.py -q -m gpu --durations=12 > gpurun_out/r3f_tests.log 2>&1
rc=$?; echo "graph + fullsize tests rc=$rc" | tee -a gpurun_out/r3f_status.txt; tail -25 gpurun_out/r3f_tests.log | cut -c1-200
