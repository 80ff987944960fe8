#!/bin/bash
# round 4, GPU call AP: per-rank MIOpen table directories: memset probe and the 2-rank rehearsal of the bench path again (one GPU, gloo)
set -o pipefail
O=gpurun_out/r4ap; mkdir -p $O
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/micro/ddp_memset_probe.py > $O/probe_2.txt 2>&1; echo "2 ranks rc=$?"; grep -E "^#|^  n=" $O/probe_2.txt | cut -c1-300
TAMTR_BENCH_ALLOW_GLOO=1 TAMTR_DIST_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_2rank.json 2> $O/bench_2rank.err; echo "2-rank bench rc=$?" | tee -a $O/status.txt; grep -E "capture|timed|graph vs" $O/bench_2rank.err | cut -c1-300; cut -c1-250 $O/bench_2rank.json
TAMTR_BENCH_ALLOW_GLOO=1 TAMTR_DIST_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus 4 --batch 4 --steps 3 --warmup 2 --no-cpu-baseline --grad-dtype bf16 > $O/bench_4rank.json 2> $O/bench_4rank.err; echo "4-rank (4 images each, bf16 buckets) rc=$?" | tee -a $O/status.txt; grep -E "capture|timed|graph vs" $O/bench_4rank.err | cut -c1-300
