#!/bin/bash
# round 4, last check at HEAD: GPU suite, smoke, default bench (with the CPU baseline), rocprofv3 kernel trace of the bench
set -o pipefail
O=gpurun_out/r4final5; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1; echo "gpu suite rc=$?" | tee -a $O/status.txt; tail -3 $O/gpu_tests.txt | cut -c1-300
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $O/status.txt; tail -2 $O/smoke.txt | cut -c1-300
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/status.txt; cut -c1-300 $O/bench.json
bash tools/prof_step.sh r04g > $O/prof.log 2>&1; echo "prof rc=$?" | tee -a $O/status.txt; head -3 gpurun_out/prof_step_r04g.txt | cut -c1-200
