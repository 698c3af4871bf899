"""Where the wall time of a Kronecker minibatch step goes on the host: Python packing vs the C call (which ends with its one stream
synchronisation)."""
import os, sys, time
import ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'zero-inflated-gp_amd'))
import zigp
from zigp import _lib
from zigp._lib import ptr
from onofftf.model import init_params, engine_params
d = np.load(os.path.join(ROOT, 'tests', 'golden', 'pptr.npz'))
Xtr, Ytr = d['Xtrain'].copy(), d['Ytrain']; Xtr[:, 2] /= 1000.0
np.random.seed(0)
pk = engine_params(init_params(Xtr, (32, 32), (32, 32), kmeans_seed=1))
eng = zigp.DenseEngine(0)
X, Y = np.ascontiguousarray(Xtr[:1000]), np.ascontiguousarray(Ytr[:1000]).reshape(-1)
for _ in range(20): eng.kron_elbo(pk, X, Y, jitter=1e-5, scale=105.28)
n = 500
t0 = time.perf_counter()
for _ in range(n): eng.kron_elbo(pk, X, Y, jitter=1e-5, scale=105.28)
t_all = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(n): s, keep, dims = eng._pack_kron(pk)
t_pack = (time.perf_counter() - t0) / n
# the bare C call with pre-packed arguments and pre-allocated gradient arrays
s, keep, dims = eng._pack_kron(pk)
gs = _lib.zigp_kron_grads(); hold = []
for tag in ('f', 'g'):
    Z0, Z1, l0, l1, um, us = keep[tag]
    for name, a in (('Z0', Z0), ('Z1', Z1), ('ell0', l0), ('ell1', l1)):
        z = np.zeros_like(a); hold.append(z); setattr(gs, name + tag, ptr(z))
    for name, a in (('u_%sm' % tag, um), ('u_%ss_sqrt' % tag, us)):
        z = np.zeros_like(a); hold.append(z); setattr(gs, name, ptr(z))
ed, kl = C.c_double(0), C.c_double(0)
t0 = time.perf_counter()
for _ in range(n):
    eng.lib.zigp_kron_elbo(eng.ctx, C.byref(s), ptr(X), ptr(Y), 1000, 1e-5, 105.28, 0.0, 0.0, 1, C.byref(ed), C.byref(kl), C.byref(gs), None)
t_c = (time.perf_counter() - t0) / n
print('minibatch step: python call %.1f us = C call %.1f us + packing %.1f us + result dicts / checks %.1f us' % (t_all * 1e6, t_c * 1e6, t_pack * 1e6, (t_all - t_c - t_pack) * 1e6))
st = eng.kron_stepper(pk)
for _ in range(20): st(pk, X, Y, jitter=1e-5, scale=105.28)
t0 = time.perf_counter()
for _ in range(n): st(pk, X, Y, jitter=1e-5, scale=105.28)
t_st = (time.perf_counter() - t0) / n
print('prepared step (DenseEngine.kron_stepper): %.1f us = C call %.1f us + %.1f us of Python' % (t_st * 1e6, t_c * 1e6, (t_st - t_c) * 1e6))
