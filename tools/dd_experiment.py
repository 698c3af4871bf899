"""Which stage of the factored Kronecker reverse pass needs more than float64?  
CPU experiment, mpmath with per-stage working precision: 53 bits emulates float64, 106 double-double.
Latent f only, gradient w.r.t. the spatial inducing inputs Z0 at the pptr init on the 32 x 32 grid (cond K_s ~ 5e7)."""
import os, sys
import numpy as np
import mpmath as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for q in ('zero-inflated-gp_amd', 'oracle', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, q))


def obj(a):
    a = np.asarray(a, dtype=np.float64)
    o = np.empty(a.shape, dtype=object)
    for i in np.ndindex(a.shape):
        o[i] = mp.mpf(float(a[i]))
    return o


def rnd(a, prec):
    with mp.workprec(prec):
        o = np.empty(a.shape, dtype=object)
        for i in np.ndindex(a.shape):
            o[i] = +a[i]
        return o


def tofloat(a):
    return np.array([float(v) for v in a.reshape(-1)]).reshape(a.shape)


def kmat(A, B, ell, var):
    o = np.empty((A.shape[0], B.shape[0]), dtype=object)
    for i in range(A.shape[0]):
        for j in range(B.shape[0]):
            r2 = mp.mpf(0)
            for d in range(A.shape[1]):
                t = (A[i, d] - B[j, d]) / ell[d]
                r2 += t * t
            o[i, j] = var * mp.exp(-r2 / 2)
    return o


def inv(K):
    M = mp.matrix(K.tolist())
    P = M ** -1
    return np.array(P.tolist(), dtype=object).reshape(K.shape)


def run(X, Z0, Z1, ell0, ell1, var0, var1, u, s, gm, gv, jit, M1o_klcoef, prec, with_kl=True):
    """prec: dict stage -> bits.  stages: K (factor build), P (inverse), lat (Alpha, T), fwd (point forward), bwd (point backward +
    sums over points), fin (finish).  gm, gv: float64 point-wise cotangents (fixed inputs here).  Returns dZ0 (float)."""
    hand = prec.get('hand', 53)   # precision the point stage sees P / Alpha in
    M0, M1 = Z0.shape[0], Z1.shape[0]
    X0, X1 = obj(X[:, :2]), obj(X[:, 2:])
    Z0o, Z1o, e0, e1 = obj(Z0), obj(Z1), obj(ell0), obj(ell1)
    v0, v1 = mp.mpf(float(var0)), mp.mpf(float(var1))
    with mp.workprec(prec['K']):
        K0 = kmat(Z0o, Z0o, e0, v0); K1 = kmat(Z1o, Z1o, e1, v1)
        for i in range(M0): K0[i, i] += mp.mpf(jit)
        for i in range(M1): K1[i, i] += mp.mpf(jit)
    with mp.workprec(prec['P']):
        P0, P1 = inv(K0), inv(K1)
    U = obj(u.reshape(M0, M1)); S2 = obj(s.reshape(M0, M1)) ** 2
    with mp.workprec(prec['lat']):
        T0 = U.dot(P1); T1 = P0.dot(U); Al = P0.dot(T0)
    P0h, P1h, Alh = rnd(P0, hand), rnd(P1, hand), rnd(Al, hand)
    gmo, gvo = obj(gm), obj(gv)
    with mp.workprec(prec['fwd']):
        k0 = kmat(Z0o, X0, e0, v0); k1 = kmat(Z1o, X1, e1, v1)        # (M, n)
        a0 = P0h.dot(k0); a1 = P1h.dot(k1)
        q0 = (k0 * a0).sum(0); q1 = (k1 * a1).sum(0)
        B0 = Alh.dot(k1); C0 = S2.dot(a1 * a1)
    with mp.workprec(prec['bwd']):
        B1 = Alh.T.dot(k0); C1 = S2.T.dot(a0 * a0)
        dA0 = 2 * gvo[None, :] * a0 * C0; dA1 = 2 * gvo[None, :] * a1 * C1
        dq0 = -gvo * q1; dq1 = -gvo * q0
        dk0 = gmo[None, :] * B0 + 2 * dq0[None, :] * a0 + P0h.dot(dA0)
        E0 = dq0[None, :] * k0 + dA0
        E1 = dq1[None, :] * k1 + dA1
        dAl = (k0 * gmo[None, :]).dot(k1.T)
        dP0 = E0.dot(k0.T)
        # data part of dZ0
        dZ_data = np.empty((M0, 2), dtype=object)
        for m in range(M0):
            for d in range(2):
                dZ_data[m, d] = sum(dk0[m, n] * k0[m, n] * (X0[n, d] - Z0o[m, d]) for n in range(X.shape[0])) / (e0[d] * e0[d])
    with mp.workprec(prec.get('f1', prec['fin'])):
        dP = dP0 + dAl.dot(T0.T)
        if with_kl:
            Q = T0.dot(U.T)
            w = np.array([sum(P1[j, j] * S2[i, j] for j in range(M1)) for i in range(M0)], dtype=object)
            dP = dP - (Q + Q.T) / 4
            for i in range(M0): dP[i, i] -= w[i] / 2
        sym = (dP + dP.T) / 2
    if 'f1' in prec: sym = rnd(sym, prec['f1'])
    with mp.workprec(prec.get('f2', prec['fin'])):
        G = -(P0.dot(sym.dot(P0)))
        if with_kl:
            G = G - P0 * (mp.mpf(M1) / 2)
    if 'f2' in prec: G = rnd(G, prec['f2'])
    with mp.workprec(prec.get('f3', prec['fin'])):
        dZ_kuu = np.empty((M0, 2), dtype=object)
        for m in range(M0):
            for d in range(2):
                acc = mp.mpf(0)
                for j in range(M0):
                    kz = K0[m, j] - (mp.mpf(jit) if m == j else 0)
                    acc += 2 * G[m, j] * kz * (Z0o[j, d] - Z0o[m, d])
                dZ_kuu[m, d] = acc / (e0[d] * e0[d])
        dZ = dZ_data + dZ_kuu
    return tofloat(dZ), tofloat(dZ_data), tofloat(dZ_kuu)


def main():
    import torch
    import zigp_oracle_torch as ot
    from test_gpu_pptr_params import _pptr_params
    Xtr, Ytr, xb, yb, p = _pptr_params((32, 32))
    n, jit, scale = 100, 1e-5, 7.0
    X, Y = xb[:n], yb[:n]
    # point-wise cotangents gm_f, gv_f at the float64 forward values (fixed inputs of the experiment)
    import zigp_oracle as o
    fm, fv = o.kron_inf(X, p['Zf'], p['ell_f'], [float(np.squeeze(v)) for v in p['var_f']], p['u_fm'], p['u_fs_sqrt'], jit)
    gmn, gvn = o.kron_inf(X, p['Zg'], p['ell_g'], [float(np.squeeze(v)) for v in p['var_g']], p['u_gm'], p['u_gs_sqrt'], jit)
    t = ot._t
    fmt, fvt = t(fm).clone().requires_grad_(True), t(fv).clone().requires_grad_(True)
    e1, e2, ev = ot.probit_expectations(t(gmn), t(gvn))
    data = torch.sum(ot.variational_expectations(e1 * fmt, e2 * fvt, ev * torch.square(fmt), t(Y).reshape(-1, 1), t(p['noise']))) * scale
    data.backward()
    gm, gv = fmt.grad.numpy().reshape(-1), fvt.grad.numpy().reshape(-1)
    args = (X, p['Zf'][0], p['Zf'][1], p['ell_f'][0].reshape(-1), p['ell_f'][1].reshape(-1), float(np.squeeze(p['var_f'][0])), float(np.squeeze(p['var_f'][1])),
            p['u_fm'].reshape(-1), p['u_fs_sqrt'].reshape(-1), gm, gv, jit, None)
    hi = dict(K=200, P=200, lat=200, fwd=200, bwd=200, fin=200, hand=200)
    truth, td, tk = run(*args, prec=hi)
    iz = np.unravel_index(np.argmax(np.abs(truth)), truth.shape)
    print('truth dZ0 max entry', iz, truth[iz], ' data part', td[iz], ' Kuu part', tk[iz])
    def report(name, prec):
        got, gd, gk = run(*args, prec=prec)
        print('%-60s entry err %.2e   max-norm err %.2e' % (name, abs(got[iz] - truth[iz]) / abs(truth[iz]), np.max(np.abs(got - truth)) / np.max(np.abs(truth))), flush=True)
    b = dict(K=53, P=53, lat=53, fwd=53, bwd=53, fin=53, hand=53)
    report('all 53', b)
    report('P, lat, fin at 106 (verdict recipe)', dict(b, P=106, lat=106, fin=106))
    report('K, P, lat, fin at 106', dict(b, K=106, P=106, lat=106, fin=106))
    report('P, lat, fin 106 + point stage sees dd P (hand 106), fwd/bwd 53', dict(b, P=106, lat=106, fin=106, hand=106))
    report('P, lat, fin, fwd, bwd 106, hand 106, K 53', dict(b, P=106, lat=106, fin=106, fwd=106, bwd=106, hand=106))
    report('everything 106', dict(K=106, P=106, lat=106, fwd=106, bwd=106, fin=106, hand=106))
    report('only fwd+bwd 106 (hand 53)', dict(b, fwd=106, bwd=106))
    report('P,lat,fin,bwd 106, fwd 53, hand 53', dict(b, P=106, lat=106, fin=106, bwd=106))
    report('P,lat,fin,fwd 106, bwd 53, hand 53', dict(b, P=106, lat=106, fin=106, fwd=106))
    report('f2+f3 106, f1 53', dict(b, f1=53, f2=106, f3=106))
    report('f2 106, f1 f3 53', dict(b, f1=53, f2=106, f3=53))
    report('f3 106 only', dict(b, f1=53, f2=53, f3=106))
    report('f1+f2 106, f3 53', dict(b, f1=106, f2=106, f3=53))
    report('f1 106 only', dict(b, f1=106, f2=53, f3=53))
    return
    report('P only 106', dict(b, P=106))
    report('fin only 106', dict(b, fin=106))
    report('P + fin 106', dict(b, P=106, fin=106))
    report('P + lat 106', dict(b, P=106, lat=106))
    report('lat + fin 106', dict(b, lat=106, fin=106))
    report('P, lat, fin at 80', dict(b, P=80, lat=80, fin=80))
    report('P, lat, fin at 70', dict(b, P=70, lat=70, fin=70))


if __name__ == '__main__':
    main()
