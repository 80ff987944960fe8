#!/bin/bash
# round 3, GPU call H: VSS levels on two streams - parity (graphs / fullsize / modules suites) and A/B bench
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
killed() { [ "$1" = 124 ] || [ "$1" = 137 ]; }
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r3h_bench_streams.json 2> gpurun_out/r3h_bench_streams.err
rc=$?; echo "bench (2 streams) rc=$rc" | tee gpurun_out/r3h_status.txt; grep -E "timed|graph vs|captured|check" gpurun_out/r3h_bench_streams.err | cut -c1-260
killed $rc && exit $rc
TAMTR_VSS_STREAMS=0 timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r3h_bench_1stream.json 2> gpurun_out/r3h_bench_1stream.err
rc=$?; echo "bench (1 stream) rc=$rc" | tee -a gpurun_out/r3h_status.txt; grep -E "timed" gpurun_out/r3h_bench_1stream.err | cut -c1-200
killed $rc && exit $rc
timeout -k 10 400 python bench.py --no-cpu-baseline --static-part eager > gpurun_out/r3h_bench_streams_eager.json 2> gpurun_out/r3h_bench_streams_eager.err
rc=$?; echo "bench (2 streams, eager) rc=$rc" | tee -a gpurun_out/r3h_status.txt; grep -E "timed" gpurun_out/r3h_bench_streams_eager.err | cut -c1-200
killed $rc && exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_graphs.py tests/test_gpu_fullsize.py tests/test_gpu_modules.py -q -m gpu -x > gpurun_out/r3h_tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee -a gpurun_out/r3h_status.txt; tail -6 gpurun_out/r3h_tests.log | cut -c1-200
