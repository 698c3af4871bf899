#!/usr/bin/env python
"""Wall-clock of the secondary BASELINE.json configs (cfg2 dense N=1e5/M=512, cfg5 Kronecker pptr 32x32) on one MI355X.
Not the contract bench (that is bench.py = cfg3); numbers land in DESIGN.md."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'zero-inflated-gp_amd'))
import bench  # noqa: E402
import zigp  # noqa: E402
from onofftf.model import init_params, engine_params  # noqa: E402


def timeit(f, n=5, warm=2):
    for _ in range(warm):
        f()
    t0 = time.time()
    for _ in range(n):
        f()
    return (time.time() - t0) / n


def main():
    eng = zigp.DenseEngine(0)
    out = {}
    X, Y, p = bench.synth(100000, 512, 3)
    eng.set_data(X, Y)
    t = timeit(lambda: eng.elbo(p))
    out['cfg2_dense_N1e5_M512'] = dict(ms_per_step=t * 1e3, steps_per_s=1 / t, tflops_10M2N=10 * 512 * 512 * 1e5 / t / 1e12)
    d = np.load(os.path.join(ROOT, 'tests', 'golden', 'pptr.npz'))
    Xtr, Ytr = d['Xtrain'].copy(), d['Ytrain']
    Xtr[:, 2] /= 1000.0
    np.random.seed(0)
    pk = engine_params(init_params(Xtr, (32, 32), (32, 32), kmeans_seed=1))
    t = timeit(lambda: eng.kron_elbo(pk, Xtr, Ytr, jitter=1e-5), n=5, warm=2)
    out['cfg5_kron_pptr_32x32_fullbatch_N105280'] = dict(ms_per_step=t * 1e3, steps_per_s=1 / t, rows_per_s=105280 / t)
    xb, yb = Xtr[:1000], Ytr[:1000]
    t = timeit(lambda: eng.kron_elbo(pk, xb, yb, jitter=1e-5, scale=105.28), n=20, warm=3)
    out['cfg5_kron_pptr_32x32_minibatch1000'] = dict(ms_per_step=t * 1e3, steps_per_s=1 / t)
    pk2 = engine_params(init_params(Xtr, (10, 100), (10, 100), kmeans_seed=1))
    t = timeit(lambda: eng.kron_elbo(pk2, xb, yb, jitter=1e-5, scale=105.28), n=20, warm=3)
    out['reference_grid_10x100_minibatch1000'] = dict(ms_per_step=t * 1e3, steps_per_s=1 / t)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
