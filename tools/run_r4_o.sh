#!/bin/bash
# round 4, GPU call O: SS2D planes in bf16 (bf16 mode): kernel-form test, composite test, per-kernel micro-benchmark, A/B bench
set -o pipefail
O=gpurun_out/r4o; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "bf16_planes or xproj or cross_merge or dwconv" > $O/t_ops.txt 2>&1; echo "kernel tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t_ops.txt | cut -c1-400 | head -20
timeout -k 10 600 python3 -m pytest tests/test_gpu_modules.py -q -m gpu -k "ss2d or scan or vss" > $O/t_mod.txt 2>&1; echo "module tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t_mod.txt | cut -c1-400 | head -20
timeout -k 10 300 python3 tools/bench_kernels.py planes > $O/planes.txt 2>&1; echo "planes microbench rc=$?" | tee -a $O/status.txt; cat $O/planes.txt | cut -c1-200
TAMTR_SS2D_PLANES=f32 timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_f32.json 2> $O/bench_f32.err; grep -E "timed" $O/bench_f32.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_p16.json 2> $O/bench_p16.err; grep -E "timed|graph vs" $O/bench_p16.err | cut -c1-300
TAMTR_SS2D_PLANES=f32 timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_f32b.json 2> $O/bench_f32b.err; grep -E "timed" $O/bench_f32b.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_p16b.json 2> $O/bench_p16b.err; grep -E "timed" $O/bench_p16b.err
timeout -k 10 600 python3 -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "fp32_elementwise or bf16_rounding or hip_path or bf16_vs" > $O/t_full.txt 2>&1; echo "fullsize tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t_full.txt | cut -c1-700 | head -20
