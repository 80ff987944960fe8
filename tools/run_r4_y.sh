#!/bin/bash
# round 4, GPU call Y: deterministic bench on node-by-node graph launches; split-K slice count A/B; event-bracket field
set -o pipefail
O=gpurun_out/r4y; mkdir -p $O
TAMTR_DETERMINISTIC=1 timeout -k 10 600 python3 bench.py --no-cpu-baseline > $O/bench_deterministic.json 2> $O/bench_deterministic.err; echo "deterministic bench rc=$?" | tee -a $O/status.txt; grep -E "timed|capture" $O/bench_deterministic.err | cut -c1-200
for S in 64 16 32 64 16; do
  TAMTR_SPLITK_MAX=$S timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_s${S}_$RANDOM.json 2> $O/bench_s.err; echo "S=$S $(grep -E 'timed' $O/bench_s.err | cut -c1-120)" | tee -a $O/splitk.txt
done
