#!/bin/bash
# round 4, GPU call J: per-phase profile after the x_proj kernels; CPU-spin experiments (runtime wait knobs)
set -o pipefail
O=gpurun_out/r4j; mkdir -p $O
timeout -k 10 400 python3 tools/step_phases.py --top 14 --json $O/step_phases.json > $O/step_phases.txt 2> $O/step_phases.err; echo "phases rc=$?" | tee -a $O/status.txt; head -16 $O/step_phases.txt | cut -c1-200
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "fused_optim" > $O/t_opt.txt 2>&1; echo "fused optim rc=$?" | tee -a $O/status.txt
for cfg in "ROC_ACTIVE_WAIT_TIMEOUT=0" "HSA_ENABLE_INTERRUPT=0" "OMP_NUM_THREADS=1"; do
  echo "== $cfg" | tee -a $O/spin.txt
  env $cfg timeout -k 10 300 python3 tools/host_phases.py 2>/dev/null | tail -3 | cut -c1-400 | tee -a $O/spin.txt
done
