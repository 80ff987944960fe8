#!/bin/bash
# PMC passes over tools/scan_only.py (level $1, default 0): SQ issue / wait counters, LDS, memory traffic.  Run on the GPU box:
#   gpurun -- 'bash tools/pmc_scan.sh 0'   -> gpurun_out/pmc_scan<level>/pass*/...counter_collection.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=${1:-0}
O=gpurun_out/pmc_scan$L
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $O/pass1 -- python3 tools/scan_only.py $L > $O/log1.txt 2>&1 &&
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pass2 -- python3 tools/scan_only.py $L > $O/log2.txt 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pass3 -- python3 tools/scan_only.py $L > $O/log3.txt 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pass4 -- python3 tools/scan_only.py $L > $O/log4.txt 2>&1
python3 tools/pmc_summary.py $O
