#!/bin/bash
# round 4, GPU call P: 32x32-tile forms of the 2-D kernels on bf16 planes: exactness tests, micro-benchmark, A/B bench
set -o pipefail
O=gpurun_out/r4p; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "bf16_planes or xproj or cross_merge or dwconv" > $O/t_ops.txt 2>&1; echo "kernel tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t_ops.txt | cut -c1-400 | head -20
timeout -k 10 300 python3 tools/bench_kernels.py planes > $O/planes.txt 2>&1; echo "planes microbench rc=$?" | tee -a $O/status.txt; grep -E "dwconv_cross_fwd|cross_merge|total" $O/planes.txt | cut -c1-200
TAMTR_SS2D_PLANES=f32 timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_f32.json 2> $O/bench_f32.err; grep -E "timed" $O/bench_f32.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_p16.json 2> $O/bench_p16.err; grep -E "timed|graph vs" $O/bench_p16.err | cut -c1-300
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_p16b.json 2> $O/bench_p16b.err; grep -E "timed" $O/bench_p16b.err
