#!/bin/bash
# rocprofv3 --kernel-trace of the default bench (no cpu_baseline; extra bench arguments in $BENCH_ARGS) + steady-state per-step summary -> gpurun_out/prof_step_<tag>.txt
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_step
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_step -o st --output-format csv -- python3 bench.py --no-cpu-baseline --no-graph-check $BENCH_ARGS > gpurun_out/prof_step_bench.json 2> gpurun_out/prof_step_bench.err
ms=$(python3 -c "import json; print(json.load(open('gpurun_out/prof_step_bench.json'))['ms_per_step'])")
python3 tools/prof_summary.py gpurun_out/prof_step/st_kernel_trace.csv 10 $ms ${TOP:-70} > gpurun_out/prof_step_$tag.txt
python3 tools/gemm_launches.py gpurun_out/prof_step/st_kernel_trace.csv 10 $ms > gpurun_out/prof_step_${tag}_gemm_launches.txt
cp gpurun_out/prof_step/st_kernel_stats.csv gpurun_out/prof_step_${tag}_kernel_stats.csv
rm -rf gpurun_out/prof_step
head -75 gpurun_out/prof_step_$tag.txt
