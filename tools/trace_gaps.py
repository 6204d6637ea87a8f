#!/usr/bin/env python3
"""From a rocprofv3 kernel-trace CSV: durations of the named kernel and the idle
gaps between its consecutive launches (last 30 launches = the timed region of
bench.py), to tell kernel time from launch-to-launch time."""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r['Kernel_Name']:
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
rows.sort()
tail = rows[-30:]
dur = [(e - s) / 1e3 for s, e in tail]
gap = [(tail[i + 1][0] - tail[i][1]) / 1e3 for i in range(len(tail) - 1)]
pitch = [(tail[i + 1][0] - tail[i][0]) / 1e3 for i in range(len(tail) - 1)]
print('launches of %s: %d; last %d: duration avg %.1f us (min %.1f, max %.1f); gap to next avg %.1f us (max %.1f); '
      'start-to-start avg %.1f us' % (sys.argv[2], len(rows), len(tail), sum(dur) / len(dur), min(dur), max(dur),
                                      sum(gap) / len(gap), max(gap), sum(pitch) / len(pitch)))
alld = [(e - s) / 1e3 for s, e in rows]
n = len(alld)
for a, b in ((0, n // 4), (n // 4, n // 2), (n // 2, 3 * n // 4), (3 * n // 4, n)):
    if b > a:
        print('  launches %5d..%5d: duration avg %.1f us' % (a, b, sum(alld[a:b]) / (b - a)))
