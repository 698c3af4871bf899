"""Diagnostic: where a tile of the Kronecker backward kernel (k_kf_backward<2,2>, pptr full batch) spends its cycles -- s_memtime stamps
at the phase boundaries, wave 0.  Needs a stamp build (stamps never go into the product library):
    git apply tools/kron_stamp_build.patch && bash tools/build_variant.sh stamps -DZIGP_KF_STAMPS && git checkout zero-inflated-gp_amd/csrc
    ZIGP_LIB=zero-inflated-gp_amd/lib/libzigp_stamps.so python tools/kron_stamps.py
Round 3 (profiles/r03t_kron_stamps.txt): the two K tiles cost 8.0 k of 32 k cycles (a scalar branch and a serialized LDS read per
row and input dimension), the moment operands 4.7 k (eight loads each behind its own branch) -- both fixed in that round."""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'zero-inflated-gp_amd'))
import zigp
from onofftf.model import init_params, engine_params
d = np.load(os.path.join(ROOT, 'tests', 'golden', 'pptr.npz'))
Xtr, Ytr = d['Xtrain'].copy(), d['Ytrain']; Xtr[:, 2] /= 1000.0
np.random.seed(0)
pk = engine_params(init_params(Xtr, (32, 32), (32, 32), kmeans_seed=1))
eng = zigp.DenseEngine(0)
X, Y = np.ascontiguousarray(Xtr), np.ascontiguousarray(Ytr).reshape(-1)
eng.set_data(X, Y)
for _ in range(5): eng.kron_elbo(pk, rows=(0, len(X)), jitter=1e-5)
t0 = time.perf_counter()
for _ in range(50): eng.kron_elbo(pk, rows=(0, len(X)), jitter=1e-5)
print('full batch step %.1f us (stamp build: slower than the product)' % ((time.perf_counter() - t0) / 50 * 1e6))
out = (C.c_ulonglong * 16)()
f = eng.lib.zigp_test_kf_stamps; f.argtypes = [C.POINTER(C.c_ulonglong)]; f.restype = C.c_int
assert f(out) == 0
v = list(out)
names = ['forward products (4)', 'B1, C1', 'P dA0, P dA1', 'stores + Al, P0, P1 sums', 'moment operands + S2 sum', 't stores + moment sums', 'K tiles (exp)', 'loop head']
nt = max(v[9], 1)
for i, nm in enumerate(names): print('%-28s %8d cycles/tile' % (nm, v[i] // nt))
print('total %d cycles for %d tiles = %d / tile' % (v[8], nt, v[8] // nt))
