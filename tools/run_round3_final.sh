#!/bin/bash
# Round-3 measurement batch on the GPU box -> gpurun_out/r03_* (copied into profiles/ afterwards).  Three gpurun calls (20 minutes each at most):
#   bash tools/run_round3_final.sh 1   whole GPU suite, smoke, default bench (with cpu_baseline), the same bench under rocprofv3 --kernel-trace --stats
#   bash tools/run_round3_final.sh 2   PMC passes over the value-projection GEMM (-> gpurun_out/gemm_pmc.json = profiles/r03_gemm_pmc.json, which bench.py reads)
#   bash tools/run_round3_final.sh 3   configs[4] (1280 px, 8 images), deterministic mode, bench --gpus 2 (self-launched, gloo ranks on the one GPU:
#                                      a rehearsal of the code path), training from image files
# A step that is killed at its time limit ends the call.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
killed() { [ "$1" = 124 ] || [ "$1" = 137 ]; }
step() { name=$1; lim=$2; shift 2; timeout -k 10 $lim "$@"; rc=$?; echo "$name rc=$rc" | tee -a gpurun_out/r03_status.txt; killed $rc && exit $rc; return 0; }
if [ "$1" = 1 ]; then
  : > gpurun_out/r03_status.txt
  step tests 900 bash -c 'python -m pytest tests -m gpu -q --durations=10 > gpurun_out/r03_gpu_tests.txt 2>&1'; tail -3 gpurun_out/r03_gpu_tests.txt
  step smoke 300 bash -c 'python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" > gpurun_out/r03_smoke.txt 2>&1'; tail -1 gpurun_out/r03_smoke.txt
  step bench 600 bash -c 'python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err'; grep "^\[bench" gpurun_out/r03_bench.err | cut -c1-220; cut -c1-1800 gpurun_out/r03_bench.json
  step prof 500 bash -c 'bash tools/prof_step.sh r03 > gpurun_out/r03_prof_step.log 2>&1'; head -12 gpurun_out/prof_step_r03.txt | cut -c1-170
elif [ "$1" = 2 ]; then
  step pmc 600 bash -c 'bash tools/pmc_gemm.sh > gpurun_out/r03_gemm_pmc.txt 2>&1'; tail -8 gpurun_out/r03_gemm_pmc.txt | cut -c1-160; head -12 gpurun_out/gemm_pmc.json
  rm -rf gpurun_out/pmc_gemm   # raw counter / trace CSVs: tens of MB, summarised above (gpurun copies back at most 64 MiB)
else
  step bench1280 500 bash -c 'python bench.py --imgsz 1280 --batch 8 --no-cpu-baseline > gpurun_out/r03_bench_1280_bs8.json 2> gpurun_out/r03_bench_1280_bs8.err'; grep -E "timed|captured" gpurun_out/r03_bench_1280_bs8.err | cut -c1-200
  step benchdet 700 bash -c 'TAMTR_DETERMINISTIC=1 python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/r03_bench_deterministic.json 2> gpurun_out/r03_bench_deterministic.err'; grep -E "timed|captured|graph vs" gpurun_out/r03_bench_deterministic.err | cut -c1-260
  step bench2rank 500 bash -c 'TAMTR_BENCH_ALLOW_GLOO=1 TAMTR_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_bench_2rank_gloo.json 2> gpurun_out/r03_bench_2rank_gloo.err'; grep -E "launch|timed" gpurun_out/r03_bench_2rank_gloo.err | cut -c1-260
  step trainfiles 500 bash -c 'python tools/train.py --synthetic 640 --batch 16 --workers 14 --epochs 3 --save-dir /tmp/r03_train_run > gpurun_out/r03_train_from_files.txt 2>&1'; tail -4 gpurun_out/r03_train_from_files.txt | cut -c1-300
fi
