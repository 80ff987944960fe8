#!/bin/bash
# round 4, GPU call A: packet-capture bisect by sub-part / BLAS backend, then the bench start-up without MIOpen's naive solvers
set -o pipefail
mkdir -p gpurun_out/r4a
O=gpurun_out/r4a
run() { # name, env..., runs the bisect
  local name=$1; shift
  echo "== $name" | tee -a $O/bisect.txt
  env "$@" timeout -k 10 300 python3 tools/graph_bisect.py "$name" 2>$O/$name.err | tee -a $O/bisect.txt
}
run pc0_vss1 PACKET_CAPTURE=0 PART=vss1 &&
run pc1_vss0 PACKET_CAPTURE=1 PART=vss0 &&
run pc1_vss1 PACKET_CAPTURE=1 PART=vss1 &&
run pc1_vss2 PACKET_CAPTURE=1 PART=vss2 &&
run pc1_proj PACKET_CAPTURE=1 PART=proj &&
run pc1_vss1_rocblas PACKET_CAPTURE=1 PART=vss1 BLAS=cublas &&
run pc1_vss1_hipblaslt PACKET_CAPTURE=1 PART=vss1 BLAS=cublaslt &&
run pc1_trunk PACKET_CAPTURE=1 PART=trunk &&
echo "== bench without naive solvers" &&
MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0 \
  timeout -k 10 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_nonaive.json 2> $O/bench_nonaive.err
echo "rc=$?" | tee -a $O/status.txt
tail -5 $O/bench_nonaive.err
