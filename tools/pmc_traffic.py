#!/usr/bin/env python3
"""Turns the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_passes.sh into the
record bench.py reports as roofline.traffic: HBM bytes per launch of one kernel
= 2 * FETCH_SIZE + WRITE_SIZE (KiB; the factor 2 is MI355X_MICROARCH.md's gfx950
rule for FETCH_SIZE), stamped with the hash of the kernel sources it was
measured on and of the plan it streamed -- both as the PROFILED program printed
them (the `fingerprint` line tools/kron_one.py leaves in every pass's log), not
recomputed afterwards.  Run on the GPU box right after the passes.

    python tools/pmc_traffic.py gpurun_out/pmc_<tag> kron_pack_kernel out.json [J_time J_space problem]
"""
import csv
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

root, kernel, out = sys.argv[1:4]
J_time, J_space, problem = (int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]) if len(sys.argv) > 6 else (6, 9, 'square')
vals = {}
names = set()
for f in glob.glob(root + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if kernel in r['Kernel_Name']:
            vals.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
            names.add(r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0])
avg = {k: sum(v) / len(v) for k, v in vals.items()}
prints = set()
for f in glob.glob(root + '/p*.log'):
    for line in open(f, errors='replace'):
        if line.startswith('fingerprint '):
            prints.add(line.strip())
assert len(prints) == 1, 'the passes do not agree on one fingerprint: %r' % sorted(prints)
stamp = dict(kv.split('=') for kv in prints.pop().split()[1:])
assert stamp['source_sha'] == bench.kernel_source_sha(), 'the tree changed since the passes ran'
fetch, write = avg['FETCH_SIZE'], avg['WRITE_SIZE']
rec = {
    'kernel': kernel,
    'kernel_instances': sorted(names),
    'J_time': J_time, 'J_space': J_space, 'problem': problem,
    'source_sha': stamp['source_sha'], 'plan_sha': stamp['plan_sha'],
    'fetch_size_kib': fetch, 'write_size_kib': write,
    'hbm_bytes_per_launch': (2.0 * fetch + write) * 1024.0,
    'launches_averaged': {k: len(v) for k, v in vals.items() if k in ('FETCH_SIZE', 'WRITE_SIZE')},
    'rule': 'hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, separate --pmc passes (MI355X_MICROARCH.md, HBM section)',
}
for k in ('TCC_EA0_RDREQ_sum', 'TCC_EA0_RDREQ_128B_sum', 'TCC_EA0_RDREQ_64B_sum', 'TCC_EA0_RDREQ_32B_sum',
          'TCC_EA0_WRREQ_sum', 'TCC_EA0_WRREQ_64B_sum', 'TCC_HIT_sum', 'TCC_MISS_sum'):
    if k in avg:
        rec.setdefault('other_counters', {})[k] = avg[k]
json.dump(rec, open(out, 'w'), indent=1, sort_keys=True)
print(json.dumps(rec))
