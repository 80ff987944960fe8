#!/bin/bash
# Round-2 measurement batch on the GPU box -> gpurun_out/r02_* (copied into profiles/ afterwards):
#   full GPU suite, smoke, default bench (with cpu_baseline), the same bench under rocprofv3 --kernel-trace --stats with its
#   steady-state per-kernel table, kernel micro-benchmarks, PMC passes over the scan backward (level 0).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r02_gpu_tests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_gpu_tests.txt
tail -3 gpurun_out/r02_gpu_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02_smoke.txt 2>&1; tail -1 gpurun_out/r02_smoke.txt
timeout -k 10 600 python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; tail -2 gpurun_out/r02_bench.err; cat gpurun_out/r02_bench.json
bash tools/prof_step.sh r02 > gpurun_out/r02_prof_step.log 2>&1
timeout -k 10 500 python tools/bench_kernels.py all > gpurun_out/r02_kernels_microbench.txt 2>&1; tail -25 gpurun_out/r02_kernels_microbench.txt
bash tools/pmc_scan.sh 0 > gpurun_out/r02_scan_pmc.txt 2>&1; head -24 gpurun_out/r02_scan_pmc.txt
