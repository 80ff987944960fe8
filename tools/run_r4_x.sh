#!/bin/bash
# round 4, GPU call X: the other modes at HEAD - deterministic bench, training from image files
set -o pipefail
O=gpurun_out/r4x; mkdir -p $O
TAMTR_DETERMINISTIC=1 timeout -k 10 600 python3 bench.py --no-cpu-baseline > $O/bench_deterministic.json 2> $O/bench_deterministic.err; echo "deterministic bench rc=$?" | tee -a $O/status.txt; cut -c1-200 $O/bench_deterministic.json
timeout -k 10 500 python3 tools/train.py --synthetic 640 --batch 16 --workers 14 --epochs 3 --save-dir /tmp/r04_train_run > $O/train_from_files.txt 2>&1; echo "train from files rc=$?" | tee -a $O/status.txt; tail -5 $O/train_from_files.txt | cut -c1-250
