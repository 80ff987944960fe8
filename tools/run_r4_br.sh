#!/bin/bash
# round 4, call BR: kernel traces with and without the paired value_proj + MSDA node
O=gpurun_out/r4br; mkdir -p $O
export TAMTR_VALUE_BIAS=colsum
TOP=400 bash tools/prof_step.sh r04br_off > $O/prof_off.log 2>&1
unset TAMTR_VALUE_BIAS
TOP=400 bash tools/prof_step.sh r04br_on > $O/prof_on.log 2>&1
python3 tools/prof_diff.py gpurun_out/prof_step_r04br_off.txt gpurun_out/prof_step_r04br_on.txt 14
