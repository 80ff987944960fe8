#!/bin/bash
# rocprofv3 kernel stats of tools/scan_only.py at the three MEH levels -> gpurun_out/prof_scan_levels.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_scanL; : > gpurun_out/prof_scan_levels.txt
for l in 0 1 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_scanL/$l -- python3 tools/scan_only.py $l > /dev/null 2>&1
  echo "== level $l" >> gpurun_out/prof_scan_levels.txt
  python3 - "$l" >> gpurun_out/prof_scan_levels.txt <<'PY'
import csv, glob, sys
f = glob.glob(f'gpurun_out/prof_scanL/{sys.argv[1]}/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:7]:
    print(f"{r['Name'].replace('(anonymous namespace)::','')[:64]:64s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f} %")
PY
done
cat gpurun_out/prof_scan_levels.txt
