#!/bin/bash
# round 4, GPU call Q: tile selection check (kernel tests), cast census of the eager step
set -o pipefail
O=gpurun_out/r4q; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "bf16_planes or cross_merge or dwconv" > $O/t_ops.txt 2>&1; echo "kernel tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t_ops.txt | cut -c1-400 | head
timeout -k 10 400 python3 tools/cast_census.py > $O/cast_census.txt 2> $O/cast_census.err; echo "cast census rc=$?" | tee -a $O/status.txt; head -60 $O/cast_census.txt | cut -c1-220
