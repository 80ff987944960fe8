#!/bin/bash
# round 4, GPU call V: fused box refinement: test, A/B bench; per-phase profile and host phases at HEAD
set -o pipefail
O=gpurun_out/r4v; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_modules.py -q -m gpu -k "box_refine or text_decoder or decoder_layer or meh_head or weight_shadows or fused_optim" > $O/t.txt 2>&1; echo "tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t.txt | cut -c1-400 | head
TAMTR_BOX_REFINE=torch timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_off.json 2> $O/bench_off.err; grep -E "timed" $O/bench_off.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_on.json 2> $O/bench_on.err; grep -E "timed" $O/bench_on.err
TAMTR_BOX_REFINE=torch timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_off2.json 2> $O/bench_off2.err; grep -E "timed" $O/bench_off2.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_on2.json 2> $O/bench_on2.err; grep -E "timed" $O/bench_on2.err
timeout -k 10 400 python3 tools/step_phases.py --top 30 --json $O/step_phases.json > $O/step_phases.txt 2> $O/step_phases.err; echo "step_phases rc=$?" | tee -a $O/status.txt; head -16 $O/step_phases.txt | cut -c1-200
timeout -k 10 300 python3 tools/host_phases.py > $O/host.txt 2> $O/host.err; tail -4 $O/host.txt | cut -c1-300
