#!/bin/bash
# PMC passes over tools/gemm_only.py (the value-projection GEMM, M = 537600, N = K = 512): MFMA / wait counters, HBM traffic.
#   gpurun -- 'bash tools/pmc_gemm.sh'   -> gpurun_out/pmc_gemm/...; summary on stdout
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_gemm
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $O/pass1 -- python3 tools/gemm_only.py > $O/log1.txt 2>&1 &&
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pass2 -- python3 tools/gemm_only.py > $O/log2.txt 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pass3 -- python3 tools/gemm_only.py > $O/log3.txt 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pass4 -- python3 tools/gemm_only.py > $O/log4.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/gemm_only.py > $O/log5.txt 2>&1
python3 tools/pmc_summary.py $O linear_bf16
python3 tools/pmc_gemm_json.py $O > gpurun_out/gemm_pmc.json
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/pmc_gemm/trace/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:4]:
    print(f"{r['Name'].replace('(anonymous namespace)::','')[:70]:70s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
