#!/bin/bash
# round 3, GPU call E: deterministic mode on an NCHW trunk (MIOpen has no deterministic non-naive NHWC bf16 solvers); start-up breakdown of the bench
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
killed() { [ "$1" = 124 ] || [ "$1" = 137 ]; }
TAMTR_DETERMINISTIC=1 TAMTR_CHANNELS_LAST=0 timeout -k 10 420 python3 tools/step_profile.py > gpurun_out/r3e_det_nchw_profile.txt 2>&1
rc=$?; echo "det nchw profile rc=$rc" | tee gpurun_out/r3e_status.txt; grep -E "^step|^profiled|^deterministic" gpurun_out/r3e_det_nchw_profile.txt; grep -A12 "by device" gpurun_out/r3e_det_nchw_profile.txt | cut -c1-200
killed $rc && exit $rc
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r3e_bench.json 2> gpurun_out/r3e_bench.err
rc=$?; echo "bench rc=$rc" | tee -a gpurun_out/r3e_status.txt; grep "^\[bench" gpurun_out/r3e_bench.err | cut -c1-250
