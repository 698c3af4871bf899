"""Minibatch Kronecker step (reference grid 10 x 100, Nb = 1000): wall clock; run under rocprofv3 --kernel-trace for the kernel list."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'zero-inflated-gp_amd'))
import zigp
from onofftf.model import init_params, engine_params
d = np.load(os.path.join(ROOT, 'tests', 'golden', 'pptr.npz'))
Xtr, Ytr = d['Xtrain'].copy(), d['Ytrain']; Xtr[:, 2] /= 1000.0
np.random.seed(0)
pk = engine_params(init_params(Xtr, (10, 100), (10, 100), kmeans_seed=1))
eng = zigp.DenseEngine(0)
xb, yb = Xtr[:1000], Ytr[:1000]
for _ in range(5): eng.kron_elbo(pk, xb, yb, jitter=1e-5, scale=105.28)
t0 = time.time()
for _ in range(50): eng.kron_elbo(pk, xb, yb, jitter=1e-5, scale=105.28)
print('minibatch step %.3f ms' % ((time.time() - t0) / 50 * 1e3))
