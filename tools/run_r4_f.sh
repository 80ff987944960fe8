#!/bin/bash
# round 4, GPU call F: packet capture ON by default (memset-node guard), fused optimizer step: new kernels' tests, bench, full suite
set -o pipefail
O=gpurun_out/r4f; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "conv1x1 or fused_optim or slab_sum" > $O/t_new.txt 2>&1; echo "new kernel tests rc=$?" | tee -a $O/status.txt; tail -12 $O/t_new.txt | cut -c1-250
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/status.txt
grep -E "forward on|capture|timed|graph vs|eager" $O/bench.err | cut -c1-300
timeout -k 10 300 python3 tools/host_phases.py > $O/host_default.txt 2> $O/host_default.err; cat $O/host_default.txt
S=$(date +%s)
timeout -k 10 1000 python3 -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1
echo "tests rc=$? wall=$(( $(date +%s) - S )) s" | tee -a $O/status.txt
tail -12 $O/gpu_tests.txt | cut -c1-300
