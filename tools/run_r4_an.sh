#!/bin/bash
# round 4, GPU call AN: 2-rank rehearsal of the bench path on ONE GPU over gloo (the code path of --gpus N: self-launch, reducer, fused optimizer
# with weight copies, graph replay); also with bf16 gradient buckets.  Not a measurement.
set -o pipefail
O=gpurun_out/r4an; mkdir -p $O
TAMTR_BENCH_ALLOW_GLOO=1 TAMTR_DIST_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_2rank.json 2> $O/bench_2rank.err; echo "2-rank rc=$?" | tee -a $O/status.txt; grep -E "launch|timed|graph vs" $O/bench_2rank.err | cut -c1-300; cut -c1-300 $O/bench_2rank.json
TAMTR_BENCH_ALLOW_GLOO=1 TAMTR_DIST_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --grad-dtype bf16 > $O/bench_2rank_bf16.json 2> $O/bench_2rank_bf16.err; echo "2-rank bf16 buckets rc=$?" | tee -a $O/status.txt; grep -E "timed" $O/bench_2rank_bf16.err | cut -c1-200
