#!/bin/bash
# round 4, GPU call R: 1x1 convolution weight gradients straight to the fp32 masters: test, graph tests, A/B bench
set -o pipefail
O=gpurun_out/r4r; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "conv1x1 or bf16_planes" > $O/t_ops.txt 2>&1; echo "kernel tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t_ops.txt | cut -c1-400 | head
timeout -k 10 900 python3 -m pytest tests/test_gpu_graphs.py -q -m gpu > $O/t_graphs.txt 2>&1; echo "graph tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t_graphs.txt | cut -c1-400 | head
TAMTR_CONV1X1_MASTER=0 timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_off.json 2> $O/bench_off.err; grep -E "timed" $O/bench_off.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_on.json 2> $O/bench_on.err; grep -E "timed|graph vs" $O/bench_on.err | cut -c1-300
TAMTR_CONV1X1_MASTER=0 timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_off2.json 2> $O/bench_off2.err; grep -E "timed" $O/bench_off2.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_on2.json 2> $O/bench_on2.err; grep -E "timed" $O/bench_on2.err
