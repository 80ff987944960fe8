#!/bin/bash
# PMC passes over the proj_conv micro-benchmark (tools/bench_kernels.py projconv): MFMA busy / waits / LDS conflicts -> gpurun_out/pmc_conv.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_conv
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $O/pass1 -- python3 tools/bench_kernels.py projconv > $O/log1.txt 2>&1 &&
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d $O/pass2 -- python3 tools/bench_kernels.py projconv > $O/log2.txt 2>&1
python3 tools/pmc_summary.py $O conv3x3_cl > gpurun_out/pmc_conv.txt
rm -rf $O
cat gpurun_out/pmc_conv.txt
