#!/usr/bin/env python3
"""Every kernel of the profiled bench run in one table: launches, average duration and
share of the kernel time (rocprofv3 --kernel-trace --stats of bench.py), and -- from the
counter passes over the solve -- bytes read and written per launch at the L2's fabric side
(2 * FETCH_SIZE, WRITE_SIZE), the rate they give and the L2 hit rate.

    python tools/kernel_table.py profiles/r05_final_kernel_stats.csv profiles/r05_final_pmc_solve.txt \
        [profiles/r05_final_pmc_kron.txt] > profiles/r05_kernel_table.md
"""
import csv
import re
import sys


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return name.split('(')[0].strip()


def counters(path):
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(' '):
            cur = out.setdefault(line.strip(), {})
        else:
            m = re.match(r'\s+(\S+)\s+n=(\d+)\s+avg=(\S+)', line)
            if m and cur is not None:
                cur[m.group(1)] = float(m.group(3))
    return out


rows = list(csv.DictReader(open(sys.argv[1])))
pmc = {}
for path in sys.argv[2:]:
    for k, v in counters(path).items():
        pmc.setdefault(k, v)
total = sum(float(r['TotalDurationNs']) for r in rows)
print('| kernel | launches | average | share of kernel time | read + written per launch (fabric side) | rate under the counter pass | L2 hit rate |')
print('|---|---|---|---|---|---|---|')
for r in rows:
    share = 100.0 * float(r['TotalDurationNs']) / total
    if share < 0.25:
        continue
    name = short(r['Name'])
    c = next((v for k, v in pmc.items() if k.startswith(name[:70])), None)
    traffic = rate = hit = ''
    if c and 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        rd, wr = 2.0 * c['FETCH_SIZE'] * 1024 / 1e6, c['WRITE_SIZE'] * 1024 / 1e6
        traffic = '%.0f + %.0f MB' % (rd, wr)
        rate = '%.1f TB/s in %.1f µs' % ((rd + wr) / c['_dur_us'] / 1e6 * 1e6 / 1e6 * 1e0, c['_dur_us']) if c.get('_dur_us') else ''
        if c.get('_dur_us'):
            rate = '%.1f TB/s (%.1f µs)' % ((rd + wr) * 1e6 / (c['_dur_us'] * 1e-6) / 1e12, c['_dur_us'])
        if 'TCC_HIT_sum' in c and 'TCC_MISS_sum' in c and c['TCC_HIT_sum'] + c['TCC_MISS_sum'] > 0:
            hit = '%.0f %%' % (100.0 * c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']))
    print('| `%s` | %s | %.1f µs | %.1f %% | %s | %s | %s |' % (name, r['Calls'], float(r['AverageNs']) / 1e3, share, traffic, rate, hit))
