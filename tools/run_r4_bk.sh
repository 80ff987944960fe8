#!/bin/bash
# round 4, call BK: after removing the two scalar-through-index assignments (host synchronisations): sync probe, tests, A/B of ops.enc_select
set -o pipefail
O=gpurun_out/r4bk; mkdir -p $O
timeout -k 10 200 python3 tools/micro/sync_probe.py 2>/dev/null | tail -4 | tee $O/sync_rows.txt
TAMTR_ENC_SELECT=dense timeout -k 10 200 python3 tools/micro/sync_probe.py 2>/dev/null | tail -4 | tee $O/sync_dense.txt
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "enc_select or fanout or zero_rows" > $O/t_ops.txt 2>&1; echo "ops tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t_ops.txt | head | cut -c1-300
for i in 1 2; do
TAMTR_ENC_SELECT=dense timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_dense$i.json 2> $O/bench_dense$i.err; grep -E "timed" $O/bench_dense$i.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_rows$i.json 2> $O/bench_rows$i.err; grep -E "timed" $O/bench_rows$i.err
done
