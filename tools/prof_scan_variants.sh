#!/bin/bash
# usage: bash tools/prof_scan_variants.sh v1 v2 ...: rocprofv3 --kernel-trace --stats of tools/bench_kernels.py scan per variant library (scan kernels only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  export TAMTR_HIP_LIB=$GRAFT_REPO_ROOT/tam-tr_amd/csrc/variants/libtamtr_$v.so
  rm -rf gpurun_out/prof_sv
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_sv -o pm --output-format csv -- python3 tools/bench_kernels.py scan > gpurun_out/prof_sv_$v.log 2>&1
  echo "== $v"; grep "^scan" gpurun_out/prof_sv_$v.log
  python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/prof_sv/pm_kernel_stats.csv')):
    if any(k in r['Name'] for k in ('selscan', 'dtproj_gdtr', 'slab_sum')):
        print(f"  {r['Name'].replace('(anonymous namespace)::','')[:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}  max {float(r['MaxNs'])/1e3:9.1f}")
PY
  rm -rf gpurun_out/prof_sv
done
