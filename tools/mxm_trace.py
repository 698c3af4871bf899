"""Kernel timeline of one M x M stage (forward + backward, no data rows) at M = 1024: run under `rocprofv3 --kernel-trace`, then
`python tools/mxm_trace.py <kernel_trace.csv>` prints the launches of the last call per stream with their gaps."""
import sys, csv
if len(sys.argv) > 1:
    rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
    # last call = everything after the last big gap (> 1 ms)
    cut = 0
    for i in range(1, len(rows)):
        if int(rows[i]['Start_Timestamp']) - int(rows[i - 1]['End_Timestamp']) > 1000000: cut = i
    rows = rows[cut:]
    t0 = int(rows[0]['Start_Timestamp'])
    agg = {}
    for r in rows:
        n = r['Kernel_Name'].split('(')[0][-70:]
        a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    print('span %.1f us, %d launches' % ((int(rows[-1]['End_Timestamp']) - t0) / 1e3, len(rows)))
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]): print('%6.1f us %4d x  %s' % (a[1], a[0], n))
else:
    sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/zero-inflated-gp_amd')
    import time, bench, zigp
    X, Y, p = bench.synth(4096, 1024, 3)
    e = zigp.DenseEngine(0); e.set_data(X, Y)
    for _ in range(3): e.elbo(p, rows=(0, 0))
    time.sleep(0.01)
    e.elbo(p, rows=(0, 0))
