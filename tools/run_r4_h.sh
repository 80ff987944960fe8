#!/bin/bash
# round 4, GPU call H: staged bf16 rows incl. the HIP path on an fp32 trunk; reworked bf16 tests; decoder cast group A/B
set -o pipefail
O=gpurun_out/r4h; mkdir -p $O
TAMTR_DETERMINISTIC=1 timeout -k 10 500 python3 tools/bf16_attribution.py --out $O/bf16_attribution.json > $O/bf16_attribution.stdout 2> $O/bf16_attribution.err
echo "attribution rc=$?" | tee -a $O/status.txt; grep "^\[attr\]" $O/bf16_attribution.err | cut -c1-330
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_graphs.py tests/test_gpu_ops.py -q -m gpu -k "bf16 or packet_capture or fused_optim or config4 or conv1x1" > $O/t_sel.txt 2>&1
echo "selected tests rc=$?" | tee -a $O/status.txt; tail -8 $O/t_sel.txt | cut -c1-300
TAMTR_DECODER_CAST_GROUP=0 timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_nocast.json 2> $O/bench_nocast.err; grep -E "timed" $O/bench_nocast.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_cast.json 2> $O/bench_cast.err; grep -E "timed|graph vs" $O/bench_cast.err | cut -c1-300
TAMTR_DECODER_CAST_GROUP=0 timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_nocast2.json 2> $O/bench_nocast2.err; grep -E "timed" $O/bench_nocast2.err
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 20 > $O/bench_cast2.json 2> $O/bench_cast2.err; grep -E "timed" $O/bench_cast2.err
