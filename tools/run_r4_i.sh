#!/bin/bash
# round 4, GPU call I: x_proj kernels (csrc/xproj.hip): kernel tests, module tests of the SS2D core, A/B bench against the torch-op form
set -o pipefail
O=gpurun_out/r4i; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "xproj or fused_optim" > $O/t_xproj.txt 2>&1; echo "xproj kernel tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t_xproj.txt | cut -c1-300 | head -20
timeout -k 10 600 python3 -m pytest tests/test_gpu_modules.py tests/test_gpu_graphs.py -q -m gpu -k "ss2d or vss or VSS or scan or replay" > $O/t_vss.txt 2>&1; echo "vss module tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t_vss.txt | cut -c1-300 | head -20
TAMTR_XPROJ=torch timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_torch.json 2> $O/bench_torch.err; grep -E "timed" $O/bench_torch.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_own.json 2> $O/bench_own.err; grep -E "timed|graph vs" $O/bench_own.err | cut -c1-300
TAMTR_XPROJ=torch timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_torch2.json 2> $O/bench_torch2.err; grep -E "timed" $O/bench_torch2.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_own2.json 2> $O/bench_own2.err; grep -E "timed" $O/bench_own2.err
timeout -k 10 300 python3 tools/host_phases.py > $O/host_default.txt 2> $O/host_default.err; cat $O/host_default.txt | cut -c1-400
