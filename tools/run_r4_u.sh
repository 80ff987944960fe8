#!/bin/bash
# round 4, GPU call U: the whole GPU suite at HEAD
set -o pipefail
O=gpurun_out/r4u; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -q -m gpu > $O/t_all.txt 2>&1; echo "gpu suite rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed|^FAILED" $O/t_all.txt | cut -c1-600 | head -20
