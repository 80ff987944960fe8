#!/bin/bash
# round 4, GPU call AU: final verification with immediate-mode tables: GPU suite, smoke, default bench (with CPU baseline), configs[4], deterministic, 2- and 4-rank rehearsals
set -o pipefail
O=${O:-gpurun_out/r4az}; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1; echo "gpu suite rc=$?" | tee -a $O/status.txt; tail -2 $O/gpu_tests.txt | cut -c1-200
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $O/status.txt
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/status.txt; grep -E "timed|capture" $O/bench.err | cut -c1-200
timeout -k 10 600 python3 bench.py --imgsz 1280 --batch 8 --no-cpu-baseline > $O/bench_1280_bs8.json 2> $O/bench_1280_bs8.err; echo "bench 1280 rc=$?" | tee -a $O/status.txt; grep -E "timed" $O/bench_1280_bs8.err | cut -c1-200
TAMTR_DETERMINISTIC=1 timeout -k 10 600 python3 bench.py --no-cpu-baseline > $O/bench_deterministic.json 2> $O/bench_deterministic.err; echo "deterministic rc=$?" | tee -a $O/status.txt; grep -E "timed" $O/bench_deterministic.err | cut -c1-200
TAMTR_BENCH_ALLOW_GLOO=1 TAMTR_DIST_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_2rank.json 2> $O/bench_2rank.err; echo "2-rank rehearsal rc=$?" | tee -a $O/status.txt; grep -E "capture|timed|graph vs" $O/bench_2rank.err | cut -c1-250
TAMTR_BENCH_ALLOW_GLOO=1 TAMTR_DIST_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus 4 --batch 4 --steps 3 --warmup 2 --no-cpu-baseline --grad-dtype bf16 > $O/bench_4rank.json 2> $O/bench_4rank.err; echo "4-rank rehearsal rc=$?" | tee -a $O/status.txt; grep -E "capture|timed|graph vs" $O/bench_4rank.err | cut -c1-250
