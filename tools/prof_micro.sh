#!/bin/bash
# rocprofv3 --kernel-trace --stats over one micro-benchmark of tools/bench_kernels.py ($1) -> gpurun_out/prof_micro_$1.csv (+ the bench's own lines)
K=${1:-projconv}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_micro
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_micro -o pm --output-format csv -- python3 tools/bench_kernels.py $K > gpurun_out/prof_micro_$K.log 2>&1
grep -v amdgpu.ids gpurun_out/prof_micro_$K.log
cp gpurun_out/prof_micro/pm_kernel_stats.csv gpurun_out/prof_micro_$K.csv
rm -rf gpurun_out/prof_micro
head -8 gpurun_out/prof_micro_$K.csv | cut -c1-160
