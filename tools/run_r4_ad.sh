#!/bin/bash
# round 4, GPU call AD: scan forward with the lane decay as exp2(A * sum dt): scan tests, kernel-time A/B of the two builds
set -o pipefail
O=gpurun_out/r4ad; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_modules.py tests/test_gpu_fullsize.py tests/test_gpu_ops.py -q -m gpu -k "scan or bf16_planes" > $O/t.txt 2>&1; echo "tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t.txt | cut -c1-300 | head -5
bash tools/prof_scan_variants.sh sumdt0 prod sumdt0 prod > $O/variants.txt 2>&1; grep -E "^==|selscan_fwd" $O/variants.txt | cut -c1-200
