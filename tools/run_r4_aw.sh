#!/bin/bash
# round 4, GPU call AW: loss curves bf16 vs fp32 on one fixed synthetic batch (200 steps each)
set -o pipefail
O=gpurun_out/r4aw; mkdir -p $O
timeout -k 10 900 python3 tools/loss_curve.py --steps 200 --out $O/loss_curve.json > $O/loss_curve.txt 2> $O/loss_curve.err; echo "rc=$?"; grep -E "^\[" $O/loss_curve.err | cut -c1-200; head -60 $O/loss_curve.txt | cut -c1-200
