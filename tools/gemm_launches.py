#!/usr/bin/env python3
"""Per-launch durations of the W-stationary GEMM inside the timed steps of a rocprofv3 --kernel-trace CSV, grouped by position in
the step (11 launches per step with different M / K: the input projections, enc_output, the three value projections and the dX
products).  The four launches bench.py's `roofline` times (M = 537 600, N = K = 512, forward) are the ones the events bracket.
    python tools/gemm_launches.py <trace.csv> <steps> <ms_per_step>"""
import csv
import statistics
import sys

path, steps, ms = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
rows = [r for r in csv.DictReader(open(path)) if 'linear_bf16_wstat_kernel' in r['Kernel_Name']]
t1 = max(int(r['End_Timestamp']) for r in rows)
sel = sorted((r for r in rows if int(r['Start_Timestamp']) > t1 - int(steps * ms * 1e6)), key=lambda r: int(r['Start_Timestamp']))
per = len(sel) // steps
sel = sel[len(sel) - per * steps:]
print(f'{len(sel)} launches in the last {steps} steps = {per} per step; position in step: kernel<K>, mean / min / max duration (us)')
for i in range(per):
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in sel[i::per]]
    name = sel[i]['Kernel_Name']
    k = name[name.index('<'):name.index('>') + 1]
    print(f'  #{i:2d}  wstat{k:6s}  {statistics.mean(d):8.1f}  {min(d):8.1f}  {max(d):8.1f}')
