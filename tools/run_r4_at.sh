#!/bin/bash
# round 4, GPU call AT: MIOpen immediate mode on the shipped tables (no timing in the process): step time, memset census alone and with 2 concurrent ranks
set -o pipefail
O=gpurun_out/r4at; mkdir -p $O
export TAMTR_CONV_FIND=immediate
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 10 > $O/bench_imm.json 2> $O/bench_imm.err; echo "bench immediate rc=$?"; grep -E "timed|capture|forward on" $O/bench_imm.err | cut -c1-250
timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 tools/micro/ddp_memset_probe.py > $O/probe_2.txt 2>&1; echo "2 ranks, concurrent, immediate rc=$?"; grep -E "^\[rank . memset|^#|^  n=" $O/probe_2.txt | cut -c1-200
TAMTR_BENCH_ALLOW_GLOO=1 TAMTR_DIST_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_2rank.json 2> $O/bench_2rank.err; echo "2-rank bench immediate rc=$?"; grep -E "capture|timed|graph vs" $O/bench_2rank.err | cut -c1-250
unset TAMTR_CONV_FIND
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 10 > $O/bench_find.json 2> $O/bench_find.err; echo "bench find rc=$?"; grep -E "timed" $O/bench_find.err | cut -c1-250
