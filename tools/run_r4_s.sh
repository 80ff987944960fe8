#!/bin/bash
# round 4, GPU call S: per-phase profile with long kernel lists (what the decoder / loss phases still launch)
set -o pipefail
O=gpurun_out/r4s; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "conv1x1" > $O/t_ops.txt 2>&1; echo "kernel tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t_ops.txt | cut -c1-300 | head -5
timeout -k 10 400 python3 tools/step_phases.py --top 45 --json $O/step_phases.json > $O/step_phases.txt 2> $O/step_phases.err; echo "step_phases rc=$?" | tee -a $O/status.txt; head -16 $O/step_phases.txt | cut -c1-200
