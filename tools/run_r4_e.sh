#!/bin/bash
# round 4, GPU call E: 1x1 convolutions' weight gradient off MIOpen's memset solvers -> whole static part under packet capture
set -o pipefail
O=gpurun_out/r4e; mkdir -p $O
run() { local name=$1; shift; echo "== $name" | tee -a $O/bisect.txt; env "$@" timeout -k 10 300 python3 tools/graph_bisect.py "$name" 2>$O/$name.err | cut -c1-3000 | tee -a $O/bisect.txt; }
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "conv1x1" > $O/t_conv1x1.txt 2>&1; echo "conv1x1 tests rc=$?" | tee -a $O/status.txt; tail -15 $O/t_conv1x1.txt | cut -c1-250
run pc1_all PACKET_CAPTURE=1 PART=all OFF_TOL=5e-2 REPLAYS=6 &&
run pc1_trunk PACKET_CAPTURE=1 PART=trunk OFF_TOL=5e-2 &&
timeout -k 10 300 python3 tools/static_census.py > $O/census.txt 2> $O/census.err
echo "census rc=$?" | tee -a $O/status.txt; head -30 $O/census.txt | cut -c1-330
PACKET_CAPTURE=0 timeout -k 10 300 python3 tools/host_phases.py > $O/host_pc0.txt 2> $O/host_pc0.err; cat $O/host_pc0.txt
PACKET_CAPTURE=1 timeout -k 10 300 python3 tools/host_phases.py > $O/host_pc1.txt 2> $O/host_pc1.err; cat $O/host_pc1.txt
