#!/bin/bash
# round 4, GPU call AS: shipped tables re-written without naive solvers: is a run now a pure look-up?  (user copy unchanged afterwards; 2 concurrent ranks on one GPU record no memset)
set -o pipefail
O=gpurun_out/r4as; mkdir -p $O
export TAMTR_MIOPEN_DB_DIR=$PWD/$O/userdb
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 10 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; grep -E "timed|capture|forward on" $O/bench.err | cut -c1-200
for f in $O/userdb/*/*.txt; do b=$(basename $f); cmp -s $f tam-tr_amd/tuned/miopen/$b && echo "$b: unchanged by the run" || echo "$b: CHANGED by the run ($(stat -c %s $f) vs $(stat -c %s tam-tr_amd/tuned/miopen/$b) bytes)"; done
unset TAMTR_MIOPEN_DB_DIR
timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 tools/micro/ddp_memset_probe.py > $O/probe_2.txt 2>&1; echo "2 ranks, concurrent rc=$?"; grep -E "^\[rank . memset|^#|^  n=" $O/probe_2.txt | cut -c1-200
TAMTR_BENCH_ALLOW_GLOO=1 TAMTR_DIST_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_2rank.json 2> $O/bench_2rank.err; echo "2-rank bench rc=$?"; grep -E "capture|timed|graph vs" $O/bench_2rank.err | cut -c1-250
