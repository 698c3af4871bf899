"""Per-kernel totals of a rocprofv3 --kernel-trace CSV: python tools/trace_top.py <dir> <steps>"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
steps = float(sys.argv[2])
rows = list(csv.DictReader(open(f)))
t = collections.defaultdict(float); n = collections.Counter()
for r in rows:
    k = r['Kernel_Name'].split('(')[0][-58:]; t[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; n[k] += 1
print('total kernel us per step: %.1f  launches per step %.1f' % (sum(t.values()) / steps, len(rows) / steps))
for k in sorted(t, key=t.get, reverse=True)[:14]:
    print('%-60s %6.1f /step  us/step %8.1f avg %6.1f' % (k, n[k] / steps, t[k] / steps, t[k] / n[k]))
