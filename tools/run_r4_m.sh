#!/bin/bash
# round 4, GPU call M: fused loss terms / matcher cost (csrc/detrloss.hip): kernel test, loss-dependent module tests, A/B bench
set -o pipefail
O=gpurun_out/r4m; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "fused_loss or fused_optim" > $O/t_loss.txt 2>&1; echo "loss kernel tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t_loss.txt | cut -c1-300 | head -20
timeout -k 10 900 python3 -m pytest tests/test_gpu_modules.py tests/test_gpu_fullsize.py -q -m gpu -k "loss or matcher or full_model or training_step or deterministic or smoke or fit" > $O/t_model.txt 2>&1; echo "model tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t_model.txt | cut -c1-300 | head -20
TAMTR_LOSS=torch timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_torch.json 2> $O/bench_torch.err; grep -E "timed" $O/bench_torch.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_own.json 2> $O/bench_own.err; grep -E "timed|graph vs" $O/bench_own.err | cut -c1-300
TAMTR_LOSS=torch timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_torch2.json 2> $O/bench_torch2.err; grep -E "timed" $O/bench_torch2.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_own2.json 2> $O/bench_own2.err; grep -E "timed" $O/bench_own2.err
timeout -k 10 300 python3 tools/host_phases.py > $O/host.txt 2> $O/host.err; cat $O/host.txt | cut -c1-300
