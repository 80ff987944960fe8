#!/bin/bash
# round 4, call BS: decoder linears' weight gradient as one product (fewer launches) against the row-sliced form, now that the phase is launch-rate-bound
O=gpurun_out/r4bs; mkdir -p $O
for i in 1 2; do
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_sliced$i.json 2> $O/bench_sliced$i.err; grep -E "timed" $O/bench_sliced$i.err
TAMTR_LINEAR_MASTER_SLICE_ROWS=1000000000 timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_whole$i.json 2> $O/bench_whole$i.err; grep -E "timed" $O/bench_whole$i.err
done
