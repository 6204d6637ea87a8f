#!/usr/bin/env python3
"""Kernel launches and device time PER ITERATION from two rocprofv3 kernel-stats
files of the same program run with n and n + extra iterations: everything that
does not scale with the iteration count (set-up, warm-up) cancels.

    tools/stats_diff.py stats_n.csv stats_n_plus_extra.csv extra
"""
import csv
import sys


def load(path):
    out = {}
    for row in csv.DictReader(open(path)):
        out[row['Name']] = (int(row['Calls']), float(row['TotalDurationNs']))
    return out


a, b, extra = load(sys.argv[1]), load(sys.argv[2]), float(sys.argv[3])
rows = []
for name, (calls, ns) in b.items():
    c0, n0 = a.get(name, (0, 0.0))
    dc, dn = (calls - c0) / extra, (ns - n0) / extra
    if abs(dc) > 1e-9:
        rows.append((dn, dc, name))
rows.sort(reverse=True)
tot_c, tot_n = sum(r[1] for r in rows), sum(r[0] for r in rows)
print('per iteration: %.1f launches, %.3f ms of kernel time' % (tot_c, tot_n / 1e6))
print('%10s %9s %9s  kernel' % ('launches', 'us total', 'us each'))
for dn, dc, name in rows:
    short = name.replace('(anonymous namespace)::', '').replace('void ', '')
    print('%10.1f %9.1f %9.2f  %s' % (dc, dn / 1e3, dn / 1e3 / dc, short[:110]))
