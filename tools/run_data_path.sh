#!/bin/bash
# GPU box: the harness fed from image files - the fit() test on the real graph, then three epochs of tools/train.py on a synthetic
# YOLO-format dataset (JPEG decode -> host kernels -> loader workers -> sync-free step -> EMA validation).
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 240 python -m pytest tests/test_gpu_modules.py -q -x -k "fit_from_image_files or engine_train" 2>&1 | tail -4 \
  && timeout -k 10 420 python tools/train.py --synthetic 192 --batch 16 --epochs 3 --workers 12 --save-dir /tmp/train_synth \
       > gpurun_out/train_synth.log 2> gpurun_out/train_synth.err \
  && cat gpurun_out/train_synth.log
