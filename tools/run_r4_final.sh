#!/bin/bash
# round 4, final measurement batch at HEAD: GPU suite, smoke, default bench (with the CPU baseline), configs[4] bench, kernel micro-benchmarks
set -o pipefail
O=gpurun_out/r4final; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1; echo "gpu suite rc=$?" | tee -a $O/status.txt; tail -3 $O/gpu_tests.txt | cut -c1-300
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $O/status.txt; tail -2 $O/smoke.txt | cut -c1-300
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/status.txt; cat $O/bench.json | cut -c1-600
timeout -k 10 600 python3 bench.py --imgsz 1280 --batch 8 --no-cpu-baseline > $O/bench_1280_bs8.json 2> $O/bench_1280_bs8.err; echo "bench 1280 rc=$?" | tee -a $O/status.txt; cat $O/bench_1280_bs8.json | cut -c1-300
timeout -k 10 600 python3 tools/bench_kernels.py all > $O/kernels_microbench.txt 2>&1; echo "microbench rc=$?" | tee -a $O/status.txt; tail -5 $O/kernels_microbench.txt | cut -c1-200
