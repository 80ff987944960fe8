#!/bin/bash
# round 4, GPU call AE: xproj_bwd_dw with contiguous 16-pixel pieces per lane: tests, kernel timing, bench
set -o pipefail
O=gpurun_out/r4ae; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_modules.py -q -m gpu -k "xproj or bf16_planes or ss2d_core or vss_block" > $O/t.txt 2>&1; echo "tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t.txt | cut -c1-300 | head -5
timeout -k 10 300 python3 tools/bench_kernels.py planes > $O/planes.txt 2>&1; grep -E "xproj_bwd_dw" $O/planes.txt | cut -c1-200
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench.json 2> $O/bench.err; grep -E "timed" $O/bench.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench2.json 2> $O/bench2.err; grep -E "timed" $O/bench2.err
