#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter CSVs per kernel (one dir per pass)."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        name = name.split('(')[0][:60]
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
        acc[name]['_dur_us'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        acc[name]['_vgpr'].append(float(r['VGPR_Count']))
        acc[name]['_lds'].append(float(r['LDS_Block_Size']))
for name, cs in sorted(acc.items()):
    if 'kron' not in name and 'gs_' not in name and 'spmm' not in name and 'wavelet' not in name:
        continue
    print(name)
    for c, v in sorted(cs.items()):
        print('   %-22s n=%-4d avg=%.6g' % (c, len(v), sum(v) / len(v)))
