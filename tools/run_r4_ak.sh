#!/bin/bash
# round 4, GPU call AK: sum_n in the step vs alone
set -o pipefail
O=gpurun_out/r4ak; mkdir -p $O
timeout -k 10 300 python3 tools/micro/sum_n_in_step.py > $O/sum_n.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/sum_n.txt | tail -8 | cut -c1-300
