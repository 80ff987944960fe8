#!/usr/bin/env python3
"""Per-kernel difference of two tools/prof_summary.py tables (ms per step, launches per step): python tools/prof_diff.py a.txt b.txt [top]"""
import re
import sys


def load(f):
    lines = open(f).read().splitlines()
    d = {}
    for l in lines[1:]:
        m = re.match(r'\s*([\d.]+) ms/step\s+[\d.]+%\s+n/step=\s*([\d.]+)\s+avg=\s*([\d.]+) us\s+(.*)', l)
        if m:
            ms, n = d.get(m.group(4), (0.0, 0.0))
            d[m.group(4)] = (ms + float(m.group(1)), n + float(m.group(2)))
    return lines[0], d


ha, a = load(sys.argv[1])
hb, b = load(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
print('a:', ha)
print('b:', hb)
rows = sorted(((b.get(k, (0, 0))[0] - a.get(k, (0, 0))[0], k) for k in set(a) | set(b)), key=lambda x: -abs(x[0]))
print(f'sum of differences (b - a): {sum(r[0] for r in rows):+.3f} ms/step')
for dlt, k in rows[:top]:
    print(f'{dlt:+7.3f} ms/step  n/step {a.get(k, (0, 0))[1]:6.1f} -> {b.get(k, (0, 0))[1]:6.1f}   {k[:120]}')
