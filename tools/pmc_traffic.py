#!/usr/bin/env python
"""Per-launch HBM traffic of each kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE collected separately, as
MI355X_MICROARCH.md section HBM prescribes).  Units: rocprofv3 reports both in KiB-like units of 1 KB?  We read the
raw counter values: FETCH_SIZE / WRITE_SIZE are in kilobytes (ROCm convention).  gfx950 correction: FETCH_SIZE under-reports
wide coalesced streaming reads by exactly 2x -> doubled here (the guide); WRITE_SIZE is exact for 16-B streaming stores.
usage: python tools/pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv > profiles/xxx.json"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name']
        tot[k] += float(r['Counter_Value'])
        cnt[k] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


f = per_kernel(sys.argv[1], 'FETCH_SIZE')
w = per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {}
for k in f:
    if 'gemm_f64_kernel' not in k and 'k_kgrad' not in k and 'k_kuf' not in k:
        continue
    fk, n = f[k]
    wk = w.get(k, (0.0, 0))[0]
    out[k] = {'launches': n, 'fetch_KB_raw_per_launch': fk, 'write_KB_raw_per_launch': wk,
              'hbm_bytes_per_launch_corrected': (2.0 * fk + wk) * 1024.0}
print(json.dumps(out, indent=1))
