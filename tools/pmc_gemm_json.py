#!/usr/bin/env python3
"""gpurun_out/pmc_gemm (tools/pmc_gemm.sh) -> the JSON record bench.py reads for roofline.traffic: HBM bytes per launch of the
value-projection GEMM from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), corrected as MI355X_MICROARCH.md prescribes for
gfx950 (FETCH_SIZE counts 64-byte units in KB of 32-byte ones for wide coalesced reads: doubled; WRITE_SIZE as is; both in KB)."""
import csv, glob, json, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc_gemm'
acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'linear_bf16_wstat' in r['Kernel_Name']:
            acc[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
avg = {k: acc[k] / cnt[k] for k in acc}
us = None
for f in glob.glob(root + '/trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'linear_bf16_wstat' in r['Name']:
            us = float(r['AverageNs']) / 1e3
M, N, K = 16 * 33600, 512, 512
hbm = avg['FETCH_SIZE'] * 2 * 1024 + avg['WRITE_SIZE'] * 1024
alg = (M * K + N * K + M * N) * 2
out = {'kernel': 'linear_bf16_wstat_kernel<512>', 'M': M, 'N': N, 'K': K, 'FETCH_SIZE_KB': avg['FETCH_SIZE'], 'WRITE_SIZE_KB': avg['WRITE_SIZE'],
       'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/gemm_only.py (tools/pmc_gemm.sh), MI355X; FETCH_SIZE doubled as '
               'MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950 (16 B per lane LDS-DMA), WRITE_SIZE taken as is',
       'hbm_bytes_per_launch': hbm, 'algorithmic_bytes': alg, 'traffic_over_algorithmic': hbm / alg, 'kernel_us_rocprof_kernel_trace': us,
       'counters': {k: avg[k] for k in sorted(avg)}}
if 'SQ_VALU_MFMA_BUSY_CYCLES' in avg and 'GRBM_GUI_ACTIVE' in avg and us:
    out['derived'] = {'effective_clock_GHz': avg['GRBM_GUI_ACTIVE'] / 8 / (us * 1e3),   # GRBM_GUI_ACTIVE is summed over the 8 XCDs
                      'mfma_busy_frac_of_elapsed': avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (avg['GRBM_GUI_ACTIVE'] * 128),
                      'wave_parked_frac': avg.get('SQ_WAIT_ANY', 0) / max(avg.get('SQ_WAVE_CYCLES', 1), 1)}
print(json.dumps(out, indent=1))
