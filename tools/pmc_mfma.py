#!/usr/bin/env python
"""MFMA pipe busy fraction per kernel from separate rocprofv3 --pmc passes (tools/pmc_mfma.sh).
SQ_VALU_MFMA_BUSY_CYCLES counts cycles (MI355X_MICROARCH.md), summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE is reported as
the sum over the 8 XCDs, so chip cycles = GRBM_GUI_ACTIVE / 8 and  busy = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 1024);
effective clock = GRBM_GUI_ACTIVE / 8 / kernel duration."""
import collections, csv, glob, json, sys

root = sys.argv[1]


def per_kernel(counter):
    f = glob.glob('%s/%s/**/*counter_collection.csv' % (root, counter), recursive=True)[0]
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == counter:
            tot[r['Kernel_Name']] += float(r['Counter_Value']); cnt[r['Kernel_Name']] += 1
    return {k: tot[k] / cnt[k] for k in tot}, cnt


def durations(counter):
    f = glob.glob('%s/%s/**/*kernel_trace.csv' % (root, counter), recursive=True)[0]
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        tot[r['Kernel_Name']] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; cnt[r['Kernel_Name']] += 1
    return {k: tot[k] / cnt[k] for k in tot}



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

mf, n = per_kernel('SQ_VALU_MFMA_BUSY_CYCLES')
gui, _ = per_kernel('GRBM_GUI_ACTIVE')
sqb, _ = per_kernel('SQ_BUSY_CYCLES')
dur = durations('GRBM_GUI_ACTIVE')
out = {}
for k in mf:
    if 'gemm_f64_kernel' not in k or k not in gui:
        continue
    cyc = gui[k] / 8.0
    out[k] = {'launches': n[k], 'avg_us_under_pmc': dur.get(k, 0.0), 'mfma_busy_frac_of_simd_cycles': mf[k] / (cyc * 1024.0),
              'clock_GHz': cyc / (dur[k] * 1e3) if dur.get(k) else None, 'raw': {'SQ_VALU_MFMA_BUSY_CYCLES': mf[k], 'GRBM_GUI_ACTIVE': gui[k], 'SQ_BUSY_CYCLES': sqb.get(k)}}
out['_meta'] = run_meta()
print(json.dumps(out, indent=1))
