#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv files under a directory per (kernel, counter), averaged per dispatch."""
import csv, glob, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', ''))[-60:]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[k][r['Counter_Name']] += 1
for k in sorted(acc, key=lambda k: -acc[k].get('SQ_WAVE_CYCLES', 0)):
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k)
    for c in sorted(acc[k]):
        print(f'    {c:28s} {acc[k][c] / cnt[k][c]:.4e}  (avg over {cnt[k][c]} dispatches)')
