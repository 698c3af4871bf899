"""Toy recipe (zero-inflated-gpflow.ipynb) continued past the notebook's 8000 L-BFGS-B iterations: where does the ELBO plateau?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'zero-inflated-gp_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from test_gpu_model import _toy_model
for seed in (int(a) for a in sys.argv[1:] or ['2', '7']):
    m, X, Y = _toy_model(seed=seed)
    tot = 0
    for leg in range(6):
        res = m.optimize(maxiter=8000)
        tot += res.nit
        print('seed %d after %6d its: ELBO %.6f (%s)' % (seed, tot, m.compute_log_likelihood(), res.message if hasattr(res, 'message') else ''), flush=True)
        if res.nit < 8000:
            break
