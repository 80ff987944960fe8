#!/bin/bash
# round 4, call BT: paired value_proj + MSDA node against the separate nodes, five alternations of 30 timed steps
O=gpurun_out/r4bt; mkdir -p $O
for i in 1 2 3 4 5; do
TAMTR_VALUE_BIAS=colsum timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-graph-check --steps 30 > $O/bench_off$i.json 2> $O/bench_off$i.err; grep -E "timed" $O/bench_off$i.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-graph-check --steps 30 > $O/bench_on$i.json 2> $O/bench_on$i.err; grep -E "timed" $O/bench_on$i.err
done
