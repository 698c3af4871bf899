"""Soak: a few thousand calls with changing shapes through every entry point; checks for crashes, NaNs and device-memory growth."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'zero-inflated-gp_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, zigp
from conftest import make_problem
from test_gpu_kron import make_kron_problem
e = zigp.DenseEngine(0)
rs = np.random.RandomState(0)
free0 = None
t0 = time.time()
for it in range(600):
    N, M, D = int(rs.randint(50, 6000)), int(rs.randint(3, 300)), int(rs.randint(1, 5))
    X, Y, p = make_problem(N, M, D, seed=it, ell=0.5)
    e.set_chunk(int(rs.choice([1024, 2048, 32768])))
    e.set_data(X, Y)
    if it % 3 == 0:
        p = dict(p, mean_b=0.1, mean_a=np.full(D, 0.05))
    e.set_overlap(bool(it % 2))
    if it % 4 == 1:      # a row-index minibatch gathered on the device (zigp_select_rows), repeats allowed
        e.select_rows(rs.randint(N, size=int(rs.randint(1, N + 1))))
    ed, kl, g = e.elbo(p)
    e.select_rows(None)
    assert np.isfinite(ed) and np.isfinite(kl) and all(np.all(np.isfinite(np.asarray(v, dtype=float))) for v in g.values())
    out = e.predict(p, X[: min(N, 700)])
    assert np.all(np.isfinite(out))
    if it % 5 == 0:
        Xk, Yk, pk = make_kron_problem(int(rs.randint(50, 1500)), int(rs.randint(2, 40)), int(rs.randint(2, 110)), seed=it)
        ek, kk, gk = e.kron_elbo(pk, Xk, Yk, jitter=1e-5, scale=3.0, f_mu=(0.1 if it % 10 == 0 else None))
        assert np.isfinite(ek) and np.isfinite(kk)
        if it % 10 == 5:     # the fit loop's prepared step on the same shapes
            st = e.kron_stepper(pk)
            e2, k2, _ = st(pk, Xk, Yk, jitter=1e-5, scale=3.0)
            e1, k1, _ = e.kron_elbo(pk, Xk, Yk, jitter=1e-5, scale=3.0)
            assert e1 == e2 and k1 == k2
        ph = {k: pk[k] for k in ('Zf', 'ell_f', 'var_f', 'u_fm', 'u_fs_sqrt', 'noise')}
        e.kron_head_elbo(ph, Xk, (Yk > 0) * 1.0, 'bernoulli')
        e.kron_head_predict(ph, Xk, 'gaussian')
    if it == 100:
        torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
torch.cuda.synchronize(); free1 = torch.cuda.mem_get_info()[0]
print('soak ok: 600 iterations in %.1f s; free device memory after iteration 100: %.1f MB, at the end: %.1f MB' % (time.time() - t0, free0 / 1e6, free1 / 1e6))
