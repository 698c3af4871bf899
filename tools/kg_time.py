import os, sys, time
ROOT = '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'zero-inflated-gp_amd'))
import bench, zigp, torch
e = zigp.DenseEngine(0)
X, Y, p = bench.synth(1000000, 1024, 3)
e.set_data_device(torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
for rows in ((0, 125000), None):
    kw = {} if rows is None else {'rows': rows}
    for ov in (0, 1):
        e.set_overlap(ov)
        for _ in range(2): e.elbo(p, **kw)
        t0 = time.time()
        for _ in range(5): e.elbo(p, **kw)
        dt = (time.time() - t0) / 5 * 1e3
        print(os.path.basename(os.environ.get('ZIGP_LIB', 'libzigp.so')), 'rows', rows, 'overlap', ov, '%.3f ms' % dt, flush=True)
    e.set_overlap(1)
    e.profile_enable(True); e.profile_sampling(1); e.profile_reset()
    for _ in range(2): e.elbo(p, **kw)
    pr = e.profile_get()
    print('   alone: ' + ' '.join('%s %.1f us/launch' % (k, v['ms'] / max(v['launches'], 1) * 1e3) for k, v in pr.items() if v['launches'] and k in ('kgrad', 'kuf_build', 'syrk', 'pointwise')), flush=True)
    e.profile_enable(False)
