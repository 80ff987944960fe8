#!/bin/bash
# round 4, GPU call Z: decoder LayerNorm / sampling-location / shared-cast changes: module + full-size tests, A/B bench, launch count under rocprofv3
set -o pipefail
O=gpurun_out/r4z; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_modules.py tests/test_gpu_fullsize.py -q -m gpu -k "decoder or msdeform or meh_head or full_model or training_step or hip_path or bf16_rounding or config4" > $O/t.txt 2>&1; echo "tests rc=$?" | tee -a $O/status.txt; grep -E "^E  |passed|failed" $O/t.txt | cut -c1-400 | head
TAMTR_DECODER_LN=torch timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_off.json 2> $O/bench_off.err; grep -E "timed" $O/bench_off.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_on.json 2> $O/bench_on.err; grep -E "timed" $O/bench_on.err
TAMTR_DECODER_LN=torch timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_off2.json 2> $O/bench_off2.err; grep -E "timed" $O/bench_off2.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_on2.json 2> $O/bench_on2.err; grep -E "timed" $O/bench_on2.err
bash tools/prof_step.sh r04z > $O/prof.log 2>&1; head -3 gpurun_out/prof_step_r04z.txt | cut -c1-200
