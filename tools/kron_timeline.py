"""Timeline of one Kronecker minibatch step (pptr, 1000 rows): `rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/kron_timeline.py run [32|100]`,
then `python tools/kron_timeline.py DIR` lists the kernels of the last step with the gaps between them (device idle time inside the step)."""
import sys, csv, glob, os
if len(sys.argv) > 1 and os.path.isdir(sys.argv[1]):
    f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    # a step ends with the last kernel before a long host gap; take the last complete step
    starts = [int(r['Start_Timestamp']) for r in rows]; ends = [int(r['End_Timestamp']) for r in rows]
    cuts = [i for i in range(1, len(rows)) if starts[i] - ends[i - 1] > 20000]
    a, b = cuts[-2], cuts[-1]
    t0 = starts[a]; prev = t0; idle = 0.0
    for r in rows[a:b]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = (s - prev) / 1e3; idle += max(gap, 0.0); prev = max(prev, e)
        print('%8.1f %8.1f  %6.1f us  (gap %5.1f)  %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, gap, r['Kernel_Name'].split('(')[0][-40:]))
    print('span %.1f us, kernels %.1f us, idle between kernels %.1f us; host gap to the next step %.1f us' %
          ((prev - t0) / 1e3, (prev - t0) / 1e3 - idle, idle, (starts[b] - prev) / 1e3))
else:
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'zero-inflated-gp_amd'))
    import zigp
    from onofftf.model import init_params, engine_params
    d = np.load(os.path.join(ROOT, 'tests', 'golden', 'pptr.npz'))
    Xtr, Ytr = d['Xtrain'].copy(), d['Ytrain']; Xtr[:, 2] /= 1000.0
    np.random.seed(0)
    grid = (10, 100) if (len(sys.argv) > 2 and sys.argv[2] == '100') else (32, 32)
    pk = engine_params(init_params(Xtr, grid, grid, kmeans_seed=1))
    eng = zigp.DenseEngine(0)
    X, Y = np.ascontiguousarray(Xtr[:1000]), np.ascontiguousarray(Ytr[:1000]).reshape(-1)
    st = eng.kron_stepper(pk)
    for _ in range(30): st(pk, X, Y, jitter=1e-5, scale=105.28)
