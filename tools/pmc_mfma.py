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
print(json.dumps(out, indent=1))
