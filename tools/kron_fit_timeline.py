"""Timeline of the device-resident Kronecker fit loop (zigp_kron_fit_steps; pptr, minibatch 1000):
`rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/kron_fit_timeline.py run [32|100]`, then
`python tools/kron_fit_timeline.py DIR` prints one step from the middle of the call (kernels, durations, gaps) and the averages over the call."""
import sys, csv, glob, os
if len(sys.argv) > 1 and os.path.isdir(sys.argv[1]):
    f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    upd = [i for i, r in enumerate(rows) if 'k_fit_update' in r['Kernel_Name']]
    # the last call's steps: consecutive k_fit_update launches (the timed call of 100 steps)
    upd = upd[-100:]
    mid = len(upd) // 2
    a, b = upd[mid - 1] + 1, upd[mid] + 1
    t0 = int(rows[a]['Start_Timestamp']); prev = int(rows[a - 1]['End_Timestamp'])
    for r in rows[a:b]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print('%8.1f %8.1f  %6.1f us  (gap %5.1f)  %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r['Kernel_Name'].split('(')[0][-40:]))
        prev = e
    span = (int(rows[upd[-1]]['End_Timestamp']) - int(rows[upd[0]]['End_Timestamp'])) / 1e3 / (len(upd) - 1)
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows[upd[0] + 1:upd[-1] + 1]) / 1e3 / (len(upd) - 1)
    print('per step over %d steps: span %.1f us, kernels %.1f us, idle between kernels %.1f us' % (len(upd) - 1, span, busy, span - busy))
    per = {}
    for r in rows[upd[0] + 1:upd[-1] + 1]:
        k = r['Kernel_Name'].split('(')[0][-40:]
        per[k] = per.get(k, 0.0) + (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 / (len(upd) - 1)
    for k, v in sorted(per.items(), key=lambda kv: -kv[1]):
        print('  %6.1f us  %s' % (v, k))
else:
    import time
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'zero-inflated-gp_amd'))
    import zigp
    from onofftf.model import init_params, KronDeviceFit
    d = np.load(os.path.join(ROOT, 'tests', 'golden', 'pptr.npz'))
    Xtr, Ytr = d['Xtrain'].copy(), d['Ytrain']; Xtr[:, 2] /= 1000.0
    np.random.seed(0)
    grid = (10, 100) if (len(sys.argv) > 2 and sys.argv[2] == '100') else (32, 32)
    eng = zigp.DenseEngine(0)
    eng.set_data(Xtr, Ytr)
    fit = KronDeviceFit(eng, init_params(Xtr, grid, grid, kmeans_seed=1))
    rbs = [int(r) for r in np.random.RandomState(3).randint(0, Xtr.shape[0] - 1000, size=100)]
    fit.steps(rbs[:20], 1000, 1e-5, 105.28)
    t0 = time.time()
    fit.steps(rbs, 1000, 1e-5, 105.28)
    print('grid %s: %.1f us per step (100 steps, one call)' % (grid, (time.time() - t0) / 100 * 1e6))
