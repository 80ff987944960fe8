#!/bin/bash
# round 4, GPU call B: memset / reduction nodes under packet capture, per-phase profile of the eager step, GPU suite with MIOpen's naive solvers off
set -o pipefail
O=gpurun_out/r4b; mkdir -p $O
PACKET_CAPTURE=0 timeout -k 10 200 python3 tools/micro/graph_memset_probe.py > $O/memset_pc0.json 2> $O/memset_pc0.err &&
PACKET_CAPTURE=1 timeout -k 10 200 python3 tools/micro/graph_memset_probe.py > $O/memset_pc1.json 2> $O/memset_pc1.err &&
cat $O/memset_pc0.json $O/memset_pc1.json &&
timeout -k 10 400 python3 tools/step_phases.py --json $O/step_phases.json > $O/step_phases.txt 2> $O/step_phases.err
echo "phases rc=$?" | tee -a $O/status.txt
head -20 $O/step_phases.txt | cut -c1-200
S=$(date +%s)
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1
echo "tests rc=$? wall=$(( $(date +%s) - S )) s" | tee -a $O/status.txt
tail -5 $O/gpu_tests.txt | cut -c1-300
