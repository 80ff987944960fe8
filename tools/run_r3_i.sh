#!/bin/bash
# round 3, GPU call I: after the XCD-aware sorted deformable backward and the bf16 d(delta) workspace: whole GPU suite, profile, bench
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
killed() { [ "$1" = 124 ] || [ "$1" = 137 ]; }
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/r3i_tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee gpurun_out/r3i_status.txt; tail -8 gpurun_out/r3i_tests.log | cut -c1-220
killed $rc && exit $rc
[ $rc != 0 ] && exit $rc
timeout -k 10 500 bash tools/prof_step.sh r03b > gpurun_out/r3i_prof.log 2>&1
rc=$?; echo "prof rc=$rc" | tee -a gpurun_out/r3i_status.txt; grep -E "window|msda|selscan|dtproj|slab_sum|fillBuffer|bfloat16_copy" gpurun_out/prof_step_r03b.txt | cut -c1-170
killed $rc && exit $rc
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r3i_bench.json 2> gpurun_out/r3i_bench.err
rc=$?; echo "bench rc=$rc" | tee -a gpurun_out/r3i_status.txt; grep -E "timed|graph vs" gpurun_out/r3i_bench.err | cut -c1-260
