#!/bin/bash
# round 4, GPU call D: packet capture after the torch reductions left the VSS path; census of what is left; host issue time by phase; gate_cl
set -o pipefail
O=gpurun_out/r4d; mkdir -p $O
run() { local name=$1; shift; echo "== $name" | tee -a $O/bisect.txt; env "$@" timeout -k 10 300 python3 tools/graph_bisect.py "$name" 2>$O/$name.err | cut -c1-3000 | tee -a $O/bisect.txt; }
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "slab_sum or colsum" > $O/t_slab.txt 2>&1; echo "slab tests rc=$?" | tee -a $O/status.txt; tail -3 $O/t_slab.txt
timeout -k 10 300 python3 -m pytest tests/test_gpu_modules.py -q -m gpu -k "gate" > $O/t_gate.txt 2>&1; echo "gate tests rc=$?" | tee -a $O/status.txt; tail -3 $O/t_gate.txt
run pc1_vss0 PACKET_CAPTURE=1 PART=vss0 TUNED=0 &&
run pc1_vss1 PACKET_CAPTURE=1 PART=vss1 TUNED=0 &&
run pc1_all PACKET_CAPTURE=1 PART=all OFF_TOL=5e-2 &&
run pc0_all PACKET_CAPTURE=0 PART=all OFF_TOL=5e-2 &&
timeout -k 10 300 python3 tools/static_census.py > $O/census.txt 2> $O/census.err
echo "census rc=$?" | tee -a $O/status.txt; head -40 $O/census.txt | cut -c1-330
PACKET_CAPTURE=0 timeout -k 10 300 python3 tools/host_phases.py > $O/host_pc0.txt 2> $O/host_pc0.err; cat $O/host_pc0.txt
PACKET_CAPTURE=1 timeout -k 10 300 python3 tools/host_phases.py > $O/host_pc1.txt 2> $O/host_pc1.err; cat $O/host_pc1.txt
timeout -k 10 300 python3 tools/bench_kernels.py gatecl > $O/gatecl.txt 2> $O/gatecl.err; cat $O/gatecl.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_graphs.py tests/test_gpu_modules.py -q -m gpu -x > $O/t_graphs_modules.txt 2>&1; echo "graphs+modules rc=$?" | tee -a $O/status.txt; tail -4 $O/t_graphs_modules.txt | cut -c1-300
