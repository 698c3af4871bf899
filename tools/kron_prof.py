"""Kronecker (cfg5) steps for a kernel trace: python tools/kron_prof.py {full|full_res|mb|mb10x100} [steps]   (run under rocprofv3 --kernel-trace --stats;
tools/trace_top.py <dir> <steps + warm-up> lists the per-kernel totals)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'zero-inflated-gp_amd'))
import zigp
from onofftf.model import init_params, engine_params
mode = sys.argv[1] if len(sys.argv) > 1 else 'full'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
d = np.load(os.path.join(ROOT, 'tests', 'golden', 'pptr.npz'))
Xtr, Ytr = d['Xtrain'].copy(), d['Ytrain']; Xtr[:, 2] /= 1000.0
np.random.seed(0)
grid = (10, 100) if mode == 'mb10x100' else (32, 32)
pk = engine_params(init_params(Xtr, grid, grid, kmeans_seed=1))
eng = zigp.DenseEngine(0)
X, Y, scale = (Xtr, Ytr, 1.0) if mode.startswith('full') else (Xtr[:1000], Ytr[:1000], 105.28)
if mode == 'full_res':      # rows of the resident data set: no host->device copy of X, Y per step
    eng.set_data(Xtr, Ytr)
    step = lambda: eng.kron_elbo(pk, rows=(0, Xtr.shape[0]), jitter=1e-5, scale=scale)
else:
    step = lambda: eng.kron_elbo(pk, X, Y, jitter=1e-5, scale=scale)
for _ in range(5): step()
t0 = time.time()
for _ in range(steps): step()
print('%s: %.3f ms per step (%d rows, grid %s)' % (mode, (time.time() - t0) / steps * 1e3, X.shape[0], grid))
