#!/bin/bash
# round 4, GPU call AG: refresh at HEAD: per-phase profile, configs[4] bench, micro-benchmark table, contrastive tests (hygiene change)
set -o pipefail
O=gpurun_out/r4ag; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "contrastive or linear_bf16 or gemm" > $O/t.txt 2>&1; echo "tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t.txt | cut -c1-300 | head -3
timeout -k 10 400 python3 tools/step_phases.py --top 12 --json $O/step_phases.json > $O/step_phases.txt 2> $O/step_phases.err; echo "step_phases rc=$?" | tee -a $O/status.txt; head -16 $O/step_phases.txt | cut -c1-200
timeout -k 10 600 python3 bench.py --imgsz 1280 --batch 8 --no-cpu-baseline > $O/bench_1280_bs8.json 2> $O/bench_1280_bs8.err; echo "bench 1280 rc=$?" | tee -a $O/status.txt; cut -c1-200 $O/bench_1280_bs8.json
timeout -k 10 600 python3 tools/bench_kernels.py all > $O/kernels_microbench.txt 2>&1; echo "microbench rc=$?" | tee -a $O/status.txt
