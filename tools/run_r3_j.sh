#!/bin/bash
# GPU call: sorted deformable backward with gout staged in LDS (tests + microbench), then the whole suite's msda-related cases
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
killed() { [ "$1" = 124 ] || [ "$1" = 137 ]; }
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_modules.py -q -m gpu -k "msdeform or colsum or MSDeform or decoder or deterministic" > gpurun_out/r3j_tests.log 2>&1
rc=$?; echo "msda tests rc=$rc" | tee gpurun_out/r3j_status.txt; tail -5 gpurun_out/r3j_tests.log | cut -c1-200
killed $rc && exit $rc
[ $rc != 0 ] && exit $rc
timeout -k 10 200 python tools/bench_kernels.py msda > gpurun_out/r3j_msda_staged.txt 2>&1; rc=$?; killed $rc && exit $rc
tail -2 gpurun_out/r3j_msda_staged.txt
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r3j_bench.json 2> gpurun_out/r3j_bench.err
rc=$?; echo "bench rc=$rc" | tee -a gpurun_out/r3j_status.txt; grep -E "timed|graph vs" gpurun_out/r3j_bench.err | cut -c1-260
