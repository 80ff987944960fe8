#!/bin/bash
# round 3, GPU call A: new parity tests (graph replay / forced choices / configs[4] kernels), packet-capture probes, default bench.
# A step that is killed at its time limit ends the call (no further GPU step after a hang).
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
killed() { [ "$1" = 124 ] || [ "$1" = 137 ]; }
timeout -k 10 900 python -m pytest tests/test_gpu_graphs.py tests/test_gpu_fullsize.py "tests/test_gpu_modules.py::test_linear_bf16_full_size" -q -m gpu -s > gpurun_out/r3a_tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee gpurun_out/r3a_status.txt; tail -5 gpurun_out/r3a_tests.log
killed $rc && exit $rc
for cfg in "PACKET_CAPTURE=0" "PACKET_CAPTURE=1" "PACKET_CAPTURE=1 HIP_FORCE_DEV_KERNARG=0" "PACKET_CAPTURE=1 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1" \
           "PACKET_CAPTURE=1 DEBUG_HIP_KERNARG_COPY_OPT=0" "PACKET_CAPTURE=1 AMD_SERIALIZE_KERNEL=3" "PACKET_CAPTURE=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1"; do
  echo "== $cfg" >> gpurun_out/r3a_probe.log
  env $cfg timeout -k 10 240 python3 tools/graph_probe.py "$cfg" >> gpurun_out/r3a_probe.log 2>gpurun_out/r3a_probe_last.err
  rc=$?
  [ $rc != 0 ] && { echo "probe rc=$rc ($cfg)" >> gpurun_out/r3a_probe.log; tail -3 gpurun_out/r3a_probe_last.err >> gpurun_out/r3a_probe.log; }
  killed $rc && exit $rc
done
tail -20 gpurun_out/r3a_probe.log
timeout -k 10 600 python bench.py > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err
rc=$?; echo "bench rc=$rc" | tee -a gpurun_out/r3a_status.txt
tail -2 gpurun_out/r3a_bench.json
