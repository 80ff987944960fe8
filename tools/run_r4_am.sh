#!/bin/bash
# round 4, GPU call AM: VERDICT r3 item 1's artefacts at HEAD: graph probe under packet capture (6 replays), host-bound check
set -o pipefail
O=gpurun_out/r4am; mkdir -p $O
PACKET_CAPTURE=1 timeout -k 10 300 python3 tools/graph_probe.py pc1 > $O/graph_probe_pc1.txt 2> $O/graph_probe_pc1.err; echo "probe pc=1 rc=$?"; tail -1 $O/graph_probe_pc1.txt | cut -c1-600
PACKET_CAPTURE=0 timeout -k 10 300 python3 tools/graph_probe.py pc0 > $O/graph_probe_pc0.txt 2> $O/graph_probe_pc0.err; echo "probe pc=0 rc=$?"; tail -1 $O/graph_probe_pc0.txt | cut -c1-600
timeout -k 10 300 python3 tools/host_bound.py > $O/host_bound.txt 2> $O/host_bound.err; echo "host_bound rc=$?"; tail -5 $O/host_bound.txt
