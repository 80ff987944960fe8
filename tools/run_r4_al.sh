#!/bin/bash
# round 4, GPU call AL: training from image files at HEAD, 6 epochs of 40 steps (fused optimizer with bf16 weight copies, fused loss, graph replay)
set -o pipefail
O=gpurun_out/r4al; mkdir -p $O
timeout -k 10 700 python3 tools/train.py --synthetic 640 --batch 16 --workers 14 --epochs 6 --save-dir /tmp/r04_train_run2 > $O/train_from_files.txt 2>&1; echo "rc=$?"; grep -E '^\{"epoch"' $O/train_from_files.txt | cut -c1-230
