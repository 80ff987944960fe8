#!/bin/bash
# PMC passes over one micro-benchmark of tools/bench_kernels.py ($1: scan|gemm|gate|msda|attn|cpam|dwconv|lsap) -> gpurun_out/pmc_$1.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
K=${1:-dwconv}
O=gpurun_out/pmc_$K
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $O/pass1 -- python3 tools/bench_kernels.py $K > $O/log1.txt 2>&1 &&
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pass2 -- python3 tools/bench_kernels.py $K > $O/log2.txt 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pass3 -- python3 tools/bench_kernels.py $K > $O/log3.txt 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pass4 -- python3 tools/bench_kernels.py $K > $O/log4.txt 2>&1
python3 tools/pmc_summary.py $O > gpurun_out/pmc_$K.txt
