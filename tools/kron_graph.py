"""What a hipGraph replay of the Kronecker minibatch step would buy (zigp_test_kron_graph): eager vs replayed ms/step."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'zero-inflated-gp_amd'))
import zigp
from zigp._lib import ptr, as_f64
from onofftf.model import init_params, engine_params
d = np.load(os.path.join(ROOT, 'tests', 'golden', 'pptr.npz'))
Xtr, Ytr = d['Xtrain'].copy(), d['Ytrain']; Xtr[:, 2] /= 1000.0
eng = zigp.DenseEngine(0)
for grid in ((10, 100), (32, 32)):
    np.random.seed(0)
    pk = engine_params(init_params(Xtr, grid, grid, kmeans_seed=1))
    s, keep, dims = eng._pack_kron(pk)
    xb, yb = as_f64(Xtr[:1000]), as_f64(Ytr[:1000]).reshape(-1)
    out = np.zeros(2)
    rc = eng.lib.zigp_test_kron_graph(eng.ctx, C.byref(s), ptr(xb), ptr(yb), 1000, 1e-5, 105.28, 200, ptr(out))
    print('grid %s: rc %d eager %.3f ms/step, graph replay %.3f ms/step' % (grid, rc, out[0], out[1]), eng.lib.zigp_last_error(eng.ctx) if rc else '')
