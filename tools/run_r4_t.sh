#!/bin/bash
# round 4, GPU call T: bf16 weight shadows kept by the optimizer kernel + decoder linears with fp32 master gradients: tests, A/B bench
set -o pipefail
O=gpurun_out/r4t; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "fused_optim or linear_master" > $O/t_ops.txt 2>&1; echo "kernel tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t_ops.txt | cut -c1-400 | head
timeout -k 10 900 python3 -m pytest tests/test_gpu_modules.py -q -m gpu -k "weight_shadows or decoder or text_decoder or meh_head or full_model or training_step or engine_train or fit_from" > $O/t_mod.txt 2>&1; echo "module tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t_mod.txt | cut -c1-400 | head
TAMTR_LINEAR_MASTER=0 timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 --weight-shadows off > $O/bench_base.json 2> $O/bench_base.err; grep -E "timed" $O/bench_base.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 --weight-shadows off > $O/bench_lin.json 2> $O/bench_lin.err; grep -E "timed" $O/bench_lin.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_sh.json 2> $O/bench_sh.err; grep -E "timed|graph vs" $O/bench_sh.err | cut -c1-300
TAMTR_LINEAR_MASTER=0 timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 --weight-shadows off > $O/bench_base2.json 2> $O/bench_base2.err; grep -E "timed" $O/bench_base2.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_sh2.json 2> $O/bench_sh2.err; grep -E "timed" $O/bench_sh2.err
