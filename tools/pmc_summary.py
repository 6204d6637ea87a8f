#!/usr/bin/env python3
"""Per-kernel sums and averages of rocprofv3 --pmc counter CSVs (one directory
per pass, as written by tools/pmc_passes.sh)."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        name = name.split('(')[0][:70]
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
        acc[name]['_dur_us'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        acc[name]['_vgpr'].append(float(r['VGPR_Count']))
        acc[name]['_lds'].append(float(r['LDS_Block_Size']))
keep = ('kron', 'gs_', 'spmm', 'wavelet', 'rows_ell', 'mg_coarse', 'axpb', 'dot_')
for name, cs in sorted(acc.items()):
    if not any(k in name for k in keep):
        continue
    print(name)
    for c, v in sorted(cs.items()):
        print('   %-28s n=%-5d avg=%-12.6g sum=%.6g' % (c, len(v), sum(v) / len(v), sum(v)))
