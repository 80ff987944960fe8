#!/bin/bash
# round 4, call BJ: kernel traces of the step with the dense query-selection graph and with ops.enc_select
set -o pipefail
O=gpurun_out/r4bj; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "enc_select" > $O/t_ops.txt 2>&1; echo "ops tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t_ops.txt | head -20 | cut -c1-300
export TAMTR_ENC_SELECT=dense
TOP=400 bash tools/prof_step.sh r04bj_dense > $O/prof_dense.log 2>&1; echo "dense rc=$?" | tee -a $O/status.txt
unset TAMTR_ENC_SELECT
TOP=400 bash tools/prof_step.sh r04bj_rows > $O/prof_rows.log 2>&1; echo "rows rc=$?" | tee -a $O/status.txt
python3 tools/prof_diff.py gpurun_out/prof_step_r04bj_dense.txt gpurun_out/prof_step_r04bj_rows.txt 40
