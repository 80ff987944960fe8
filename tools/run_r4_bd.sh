#!/bin/bash
# round 4, call BD: RCCL tests incl. the stdout contract, then kernel traces of the N = 1 step and the --rccl-solo step
set -o pipefail
O=gpurun_out/r4bd; mkdir -p $O
timeout -k 10 400 python3 -m pytest tests/test_gpu_rccl.py -x -q -m gpu > $O/rccl_tests.txt 2>&1; echo "rccl tests rc=$?" | tee -a $O/status.txt; tail -3 $O/rccl_tests.txt | cut -c1-300
TOP=400 bash tools/prof_step.sh r04bd_n1 > $O/prof_n1.log 2>&1; echo "n1 rc=$?" | tee -a $O/status.txt
BENCH_ARGS=--rccl-solo TOP=400 bash tools/prof_step.sh r04bd_solo > $O/prof_solo.log 2>&1; echo "solo rc=$?" | tee -a $O/status.txt
python3 tools/prof_diff.py gpurun_out/prof_step_r04bd_n1.txt gpurun_out/prof_step_r04bd_solo.txt 30
