#!/bin/bash
# round 4, GPU call C: memset nodes on reused pool blocks under packet capture; default bench (EMA in the step, naive solvers out of the search); full GPU suite
set -o pipefail
O=gpurun_out/r4c; mkdir -p $O
PACKET_CAPTURE=0 timeout -k 10 200 python3 tools/micro/graph_memset_probe.py > $O/memset_pc0.json 2> $O/memset_pc0.err &&
PACKET_CAPTURE=1 timeout -k 10 200 python3 tools/micro/graph_memset_probe.py > $O/memset_pc1.json 2> $O/memset_pc1.err &&
tail -n 1 $O/memset_pc0.json | cut -c1-1500 && tail -n 1 $O/memset_pc1.json | cut -c1-1500 &&
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/status.txt
grep -E "forward on|capture|timed" $O/bench.err | cut -c1-200
S=$(date +%s)
timeout -k 10 1000 python3 -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1
echo "tests rc=$? wall=$(( $(date +%s) - S )) s" | tee -a $O/status.txt
tail -8 $O/gpu_tests.txt | cut -c1-300
