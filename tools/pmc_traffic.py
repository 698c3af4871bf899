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



def run_meta():
    """provenance: hash of the kernel sources and the configuration the counters were collected on (bench.py only quotes a
    summary whose hash and (M, chunk, D) match the run it is printed with)"""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'zero-inflated-gp_amd'))
    from zigp import build as zb
    return {'csrc_hash': zb.source_hash(zb.DENSE_FILES), 'M': int(os.environ.get('PMC_M', 1024)), 'chunk': int(os.environ.get('PMC_CHUNK', 32768)),
            'D': int(os.environ.get('PMC_D', 3)), 'rows': int(os.environ.get('ROWS', 262144)),
            'command': 'bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs --profile-steps 0 --no-overlap --rows $ROWS'}

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
out['_meta'] = run_meta()
print(json.dumps(out, indent=1))
