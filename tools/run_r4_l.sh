#!/bin/bash
# round 4, GPU call L: census of memset / reduction nodes in the label-dependent part; clean kernel trace of the bench
set -o pipefail
O=gpurun_out/r4l; mkdir -p $O
timeout -k 10 300 python3 tools/dynamic_census.py > $O/dynamic_census.txt 2> $O/dynamic_census.err; echo "census rc=$?" | tee -a $O/status.txt; cut -c1-320 $O/dynamic_census.txt | head -70
bash tools/prof_step.sh r04 > $O/prof_step_head.txt 2>&1; echo "prof rc=$?" | tee -a $O/status.txt; head -30 $O/prof_step_head.txt | cut -c1-200
