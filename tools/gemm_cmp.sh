cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 tools/bench_kernels.py gemm 2>&1 | grep -v amdgpu.ids
rm -rf gpurun_out/prof_gemm2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gemm2 -- python3 tools/bench_kernels.py gemm 2>&1 | grep "linear_bf16\|hipBLASLt"
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_gemm2/*/*kernel_trace.csv')[0]
d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if 'wstat' in r['Kernel_Name']]
print('wstat launches', len(d), 'min', min(d), 'median', sorted(d)[len(d)//2], 'max', max(d), 'first 12', d[:12])
PY
