#!/bin/bash
# round 4, GPU call AQ: memset-based solvers with several ranks: torchrun alone? concurrency on the shared GPU?
set -o pipefail
O=gpurun_out/r4aq; mkdir -p $O
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 tools/micro/ddp_memset_probe.py > $O/probe_torchrun1.txt 2>&1; echo "torchrun, 1 rank rc=$?"; grep -E "^\[rank . memset|^#|^  n=" $O/probe_torchrun1.txt | cut -c1-200
PROBE_SERIAL=1 timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 tools/micro/ddp_memset_probe.py > $O/probe_serial2.txt 2>&1; echo "2 ranks, one after the other rc=$?"; grep -E "^\[rank . memset|^#|^  n=" $O/probe_serial2.txt | cut -c1-200
timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 tools/micro/ddp_memset_probe.py > $O/probe_2.txt 2>&1; echo "2 ranks, concurrent rc=$?"; grep -E "^\[rank . memset|^#|^  n=" $O/probe_2.txt | cut -c1-200
