#!/bin/bash
# round 3, GPU call D: what is slow in deterministic mode; GEMM kernel A/B; linear tests on the new kernel; bench --gpus 2 self-launch (gloo rehearsal)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
killed() { [ "$1" = 124 ] || [ "$1" = 137 ]; }
TAMTR_DETERMINISTIC=1 timeout -k 10 420 python3 tools/step_profile.py > gpurun_out/r3d_det_profile.txt 2>&1
rc=$?; echo "det profile rc=$rc" | tee gpurun_out/r3d_status.txt; tail -34 gpurun_out/r3d_det_profile.txt | cut -c1-230
killed $rc && exit $rc
for k in ws1 ws2; do
  TAMTR_GEMM=$k timeout -k 10 200 python3 tools/gemm_ab.py >> gpurun_out/r3d_gemm_ab.txt 2>gpurun_out/r3d_gemm_ab.err
  rc=$?; [ $rc != 0 ] && { echo "gemm_ab $k rc=$rc" | tee -a gpurun_out/r3d_status.txt; tail -5 gpurun_out/r3d_gemm_ab.err; }
  killed $rc && exit $rc
done
GM=1075200 TAMTR_GEMM=ws2 timeout -k 10 200 python3 tools/gemm_ab.py >> gpurun_out/r3d_gemm_ab.txt 2>>gpurun_out/r3d_gemm_ab.err
GM=1000 TAMTR_GEMM=ws2 timeout -k 10 200 python3 tools/gemm_ab.py >> gpurun_out/r3d_gemm_ab.txt 2>>gpurun_out/r3d_gemm_ab.err
cat gpurun_out/r3d_gemm_ab.txt
timeout -k 10 600 python -m pytest tests/test_gpu_modules.py -q -m gpu -k "linear" > gpurun_out/r3d_tests.log 2>&1
rc=$?; echo "linear tests rc=$rc" | tee -a gpurun_out/r3d_status.txt; tail -4 gpurun_out/r3d_tests.log
killed $rc && exit $rc
TAMTR_BENCH_ALLOW_GLOO=1 TAMTR_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3d_bench_2rank.json 2> gpurun_out/r3d_bench_2rank.err
rc=$?; echo "bench --gpus 2 (self-launched, gloo on one GPU) rc=$rc" | tee -a gpurun_out/r3d_status.txt; tail -1 gpurun_out/r3d_bench_2rank.json | cut -c1-900; tail -4 gpurun_out/r3d_bench_2rank.err | cut -c1-300
cp -r ~/.cache/tamtr_amd gpurun_out/r3d_miopen_cache 2>/dev/null; ls -la gpurun_out/r3d_miopen_cache/* 2>/dev/null | head
