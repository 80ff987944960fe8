#!/usr/bin/env python3
"""Steady-state per-step kernel breakdown from a rocprofv3 --kernel-trace CSV: python tools/prof_summary.py <trace.csv> <steps> <ms_per_step> [top]"""
import collections
import csv
import sys

path, steps, ms = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
rows = list(csv.DictReader(open(path)))
t1 = max(int(r['End_Timestamp']) for r in rows)
w = int(steps * ms * 1e6)
sel = [r for r in rows if int(r['Start_Timestamp']) > t1 - w]
agg = collections.defaultdict(lambda: [0, 0])
for r in sel:
    a = agg[r['Kernel_Name']]
    a[0] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    a[1] += 1
tot = sum(a[0] for a in agg.values())
print(f'window: last {steps} steps x {ms} ms; {len(sel)} kernels; GPU busy {tot / steps / 1e6:.1f} ms/step; {len(sel) / steps:.0f} launches/step')
for k, (d, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f'{d / steps / 1e6:8.2f} ms/step {100 * d / tot:5.1f}%  n/step={n / steps:7.1f}  avg={d / n / 1e3:9.1f} us  {k[:110]}')
