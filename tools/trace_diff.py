#!/usr/bin/env python3
"""Kernel launches and device time PER ITERATION, by kernel AND grid size (the
grid tells the multigrid level of a row-engine launch), from two rocprofv3 kernel
TRACES of the same program run with n and n + extra iterations: set-up and
warm-up cancel.

    tools/trace_diff.py trace_n.csv trace_n_plus_extra.csv extra
"""
import collections
import csv
import sys


def load(path):
    out = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        grid = r.get('Grid_Size_X') or r.get('Grid_Size') or '?'
        wg = r.get('Workgroup_Size_X') or r.get('Workgroup_Size') or '1'
        try:
            groups = int(grid) // max(int(wg), 1)
        except ValueError:
            groups = -1
        key = (r['Kernel_Name'], groups)
        out[key][0] += 1
        out[key][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    return out


a, b, extra = load(sys.argv[1]), load(sys.argv[2]), float(sys.argv[3])
rows = []
for key, (calls, ns) in b.items():
    c0, n0 = a.get(key, (0, 0.0))
    dc, dn = (calls - c0) / extra, (ns - n0) / extra
    if abs(dc) > 1e-9:
        rows.append((key[0], -key[1], dc, dn))
rows.sort()
print('per iteration: %.1f launches, %.3f ms of kernel time' % (sum(r[2] for r in rows), sum(r[3] for r in rows) / 1e6))
print('%10s %10s %9s %9s  kernel' % ('workgroups', 'launches', 'us total', 'us each'))
for name, g, dc, dn in rows:
    short = name.replace('(anonymous namespace)::', '').replace('void ', '')
    print('%10d %10.1f %9.1f %9.2f  %s' % (-g, dc, dn / 1e3, dn / 1e3 / dc, short[:100]))
