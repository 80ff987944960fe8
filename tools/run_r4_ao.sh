#!/bin/bash
# round 4, GPU call AO: memset nodes in the recorded backward with more than one rank: attribution
set -o pipefail
O=gpurun_out/r4ao; mkdir -p $O
timeout -k 10 300 python3 tools/micro/ddp_memset_probe.py > $O/probe_1.txt 2>&1; echo "plain rc=$?"; grep -E "^\[rank|^#|^  n=" $O/probe_1.txt | cut -c1-300
OMP_NUM_THREADS=1 timeout -k 10 300 python3 tools/micro/ddp_memset_probe.py > $O/probe_1_omp1.txt 2>&1; echo "OMP=1 rc=$?"; grep -E "^\[rank|^#|^  n=" $O/probe_1_omp1.txt | cut -c1-300
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/micro/ddp_memset_probe.py > $O/probe_2.txt 2>&1; echo "2 ranks rc=$?"; grep -E "^\[rank|^#|^  n=" $O/probe_2.txt | cut -c1-300
