#!/bin/bash
# round 4, GPU call N: token memory written in place (tamtr_bncl_act_seg_*): kernel test, fp32 elementwise test with diagnostics, A/B bench
set -o pipefail
O=gpurun_out/r4n; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "bn_" > $O/t_bn.txt 2>&1; echo "bn kernel tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t_bn.txt | cut -c1-400 | head -20
timeout -k 10 600 python3 -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "fp32_elementwise or bf16_rounding or hip_path" > $O/t_full.txt 2>&1; echo "fullsize tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t_full.txt | cut -c1-700 | head -20
TAMTR_BN_CAT=torch timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_cat.json 2> $O/bench_cat.err; grep -E "timed" $O/bench_cat.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_seg.json 2> $O/bench_seg.err; grep -E "timed|graph vs" $O/bench_seg.err | cut -c1-300
TAMTR_BN_CAT=torch timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_cat2.json 2> $O/bench_cat2.err; grep -E "timed" $O/bench_cat2.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_seg2.json 2> $O/bench_seg2.err; grep -E "timed" $O/bench_seg2.err
