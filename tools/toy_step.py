"""Latency of one ELBO step on the toy configuration (cfg1 geometry: N=450, D=1, M=50), value+gradient."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'zero-inflated-gp_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import zigp
from conftest import make_problem
X, Y, p = make_problem(450, 50, 1, seed=0, ell=2.0)
X = X * 10; p['Zf'] = p['Zf'] * 10; p['Zg'] = p['Zg'] * 10
e = zigp.DenseEngine(0); e.set_data(X, Y)
for _ in range(10): e.elbo(p)
t0 = time.time()
for _ in range(200): e.elbo(p)
print('toy step %.3f ms' % ((time.time() - t0) / 200 * 1e3))
