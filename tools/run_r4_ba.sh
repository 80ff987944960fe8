#!/bin/bash
# round 4, call BA: one-rank RCCL rehearsal (tests + bench --rccl-solo with fp32 and bf16 wire buckets)
O=${O:-gpurun_out/r4ba}; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_rccl.py -x -q -m gpu > $O/rccl_tests.txt 2>&1; echo "rccl tests rc=$?" | tee -a $O/status.txt
tail -3 $O/rccl_tests.txt
timeout -k 10 300 python bench.py --rccl-solo --no-cpu-baseline > $O/bench_rccl_solo_fp32.json 2> $O/bench_rccl_solo_fp32.err; echo "solo fp32 rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python bench.py --rccl-solo --grad-dtype bf16 --no-cpu-baseline > $O/bench_rccl_solo_bf16.json 2> $O/bench_rccl_solo_bf16.err; echo "solo bf16 rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; echo "n1 rc=$?" | tee -a $O/status.txt
python - <<'PY'
import json,glob,os
O=os.environ.get('O','gpurun_out/r4ba')
for f in sorted(glob.glob(O+'/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(d['ms_per_step'],2),'ms', d['config']['static_part'], d['config'].get('dist_backend'), d['host_cpu_ms_per_step']['max'])
    except Exception as e: print(f, 'ERR', e)
PY
