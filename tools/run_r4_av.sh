#!/bin/bash
# round 4, GPU call AV: the bench's other switches still work at HEAD
set -o pipefail
O=gpurun_out/r4av; mkdir -p $O
for flags in "--dtype fp32" "--static-part eager" "--optim-step torch" "--weight-shadows off" "--conv-tuning off" "--batch 4 --imgsz 320"; do
  tag=$(echo $flags | tr -d ' -' | cut -c1-24)
  timeout -k 10 500 python3 bench.py --no-cpu-baseline --steps 3 --warmup 2 $flags > $O/bench_$tag.json 2> $O/bench_$tag.err; rc=$?
  echo "bench $flags: rc=$rc $(grep -E 'timed' $O/bench_$tag.err | cut -c20-90) $(grep -E 'capture failed' $O/bench_$tag.err | cut -c20-120)" | tee -a $O/status.txt
done
