#!/bin/bash
# round 4, GPU call K: conv3x3 epilogue swizzle + cross_merge load batching (tests, micro-benchmarks, PMC), bench with the extra roofline entries, kernel trace
set -o pipefail
O=gpurun_out/r4k; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_modules.py -q -m gpu -k "conv3x3 or gate or ss2d or cross_merge or ln_gate or vss or VSS" > $O/t_sel.txt 2>&1; echo "selected tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t_sel.txt | cut -c1-300 | head
timeout -k 10 300 python3 tools/bench_kernels.py projconv gatecl > $O/micro.txt 2> $O/micro.err; cat $O/micro.txt
bash tools/pmc_conv.sh > /dev/null 2>&1; cp gpurun_out/pmc_conv.txt $O/pmc_conv.txt; grep -E "BANK_CONFLICT|ACTIVE_INST_LDS|WAVE_CYCLES|SQ_WAVES" $O/pmc_conv.txt
bash tools/pmc_kernel.sh gatecl > /dev/null 2>&1; cp gpurun_out/pmc_gatecl.txt $O/pmc_gatecl.txt; head -40 $O/pmc_gatecl.txt
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/status.txt; grep -E "timed|graph vs" $O/bench.err | cut -c1-300
bash tools/prof_step.sh r04 > $O/prof_step_head.txt 2>&1; echo "prof rc=$?" | tee -a $O/status.txt; head -45 $O/prof_step_head.txt | cut -c1-200
