"""Kernel timeline of one M x M stage (forward + backward) at M = 1024: run under `rocprofv3 --kernel-trace`, then
`python tools/mxm_trace.py <kernel_trace.csv> [list]` prints the launches of the last call by kernel (and, with `list`, in start order with
their stream / queue and the gap to the previous end on the same queue).  MXM_ROWS=n (default 0): data rows of the call -- with rows the
backward stage also runs the products that take the accumulated cotangents in (one short chunk in between)."""
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
    if len(sys.argv) > 2:
        last = {}
        for r in rows:
            q = r.get('Queue_Id', r.get('Stream_Id', '?'))
            st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            gap = (st - last[q]) / 1e3 if q in last else 0.0
            last[q] = en
            print('%8.1f  +%6.1f us  gap %6.1f  q%s  %s' % ((st - t0) / 1e3, (en - st) / 1e3, gap, q, r['Kernel_Name'].split('(')[0][-80:]))
else:
    sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/zero-inflated-gp_amd')
    import time, bench, zigp
    import os
    nr = int(os.environ.get('MXM_ROWS', '0'))
    X, Y, p = bench.synth(max(4096, nr), 1024, 3)
    e = zigp.DenseEngine(0); e.set_data(X, Y)
    for _ in range(3): e.elbo(p, rows=(0, nr))
    time.sleep(0.01)
    e.elbo(p, rows=(0, nr))
