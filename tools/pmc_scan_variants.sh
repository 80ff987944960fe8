#!/bin/bash
# usage: bash tools/pmc_scan_variants.sh LEVEL v1 v2 ...: the SQ issue / wait counters of tools/scan_only.py per variant library
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=$1; shift
for v in "$@"; do
  O=gpurun_out/pmc_scanv_$v
  rm -rf $O; mkdir -p $O
  export TAMTR_HIP_LIB=$GRAFT_REPO_ROOT/tam-tr_amd/csrc/variants/libtamtr_$v.so
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $O/pass1 -- python3 tools/scan_only.py $L > $O/log1.txt 2>&1
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_SALU --output-format csv -d $O/pass2 -- python3 tools/scan_only.py $L > $O/log2.txt 2>&1
  echo "== $v"; python3 tools/pmc_summary.py $O selscan
  rm -rf $O
done
