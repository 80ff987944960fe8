#!/bin/bash
# round 4, GPU call G: bf16 attribution (deterministic mode, oracle's choices), then the full GPU suite (packet-capture tests in their own processes)
set -o pipefail
O=gpurun_out/r4g; mkdir -p $O
TAMTR_DETERMINISTIC=1 timeout -k 10 500 python3 tools/bf16_attribution.py --out $O/bf16_attribution.json > $O/bf16_attribution.stdout 2> $O/bf16_attribution.err
echo "attribution rc=$?" | tee -a $O/status.txt; grep "^\[attr\]" $O/bf16_attribution.err | cut -c1-330
S=$(date +%s)
timeout -k 10 1100 python3 -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1
echo "tests rc=$? wall=$(( $(date +%s) - S )) s" | tee -a $O/status.txt
tail -12 $O/gpu_tests.txt | cut -c1-300
