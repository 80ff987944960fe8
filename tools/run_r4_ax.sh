#!/bin/bash
# round 4, GPU call AX: the recorded static part reads the optimizer's weight copies: graph tests, A/B bench
set -o pipefail
O=gpurun_out/r4ax; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_graphs.py -q -m gpu > $O/t.txt 2>&1; echo "graph tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t.txt | cut -c1-400 | head -6
for v in off on off on; do
  timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 --weight-shadows $v > $O/bench_$v.json 2> $O/bench.err; echo "--weight-shadows $v: $(grep -E 'timed' $O/bench.err | cut -c1-100)" | tee -a $O/ab.txt
done
