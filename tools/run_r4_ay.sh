#!/bin/bash
set -o pipefail
O=gpurun_out/r4ay; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_graphs.py -q -m gpu -k "weight_copies" > $O/t.txt 2>&1; echo "test rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t.txt | cut -c1-500 | head -6
