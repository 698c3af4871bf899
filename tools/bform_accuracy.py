"""CPU experiment (round 6): what the "B-form" would cost in accuracy.  With B = Kuu^-1 S Kuu^-1 - Kuu^-1 (symmetric, M x M) the gradient panel
is J' = B K and the predictive variance is var = sigma^2 + colsum(K o J'), the mean alpha^T K: a gradient step would need ONE full product and
one rank-N update per latent -- 3 M^2 N flops instead of the 5 M^2 N of the W-form (A1 = W K, A2 = W^T A1, J', update).  This prints, per
configuration, cond(Kuu) and the relative error of the variance of both forms against an 80-bit evaluation with iterative refinement.
Result (profiles/r06m_bform_accuracy.log): the W-form stays at 1e-12 .. 1e-8; the B-form is at 6e-11 for cfg2 (cond 3e4), 3e-8 .. 6e-8 for cfg3
(cond 2e6), 1e-4 at cond 1e7 and O(1) beyond 1e8, with a spread of 500x at equal cond -- no guard on cond(Kuu) keeps it inside the 1e-6 of the
parity bar with a margin, so the engine computes mean / var the way the reference does (GPConditional, onofftf/main.py:257-305) and pays the flops."""
import os
import sys

import numpy as np
import scipy.linalg as sl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import bench
import zigp_oracle as o
from conftest import make_problem


def study(name, Z, ellv, var, s, X, jit=1e-6):
    M = Z.shape[0]
    Kuu = o.rbf_K(Z, Z, ellv, var) + jit * np.eye(M)
    cond = np.linalg.cond(Kuu)
    L = sl.cholesky(Kuu, lower=True)
    Kuf = o.rbf_K(Z, X[:48], ellv, var)
    W = sl.solve_triangular(L, np.eye(M), lower=True)
    P = W.T @ W
    B = P @ (s[:, None] ** 2 * P) - P
    B = 0.5 * (B + B.T)
    var_b = var + np.sum(Kuf * (B @ Kuf), 0)
    Kl, kl = Kuu.astype(np.longdouble), Kuf.astype(np.longdouble)

    def solve_ld(b):          # Kuu^-1 b in 80-bit arithmetic: float64 factor + iterative refinement
        x = sl.cho_solve((L, True), b.astype(np.float64)).astype(np.longdouble)
        for _ in range(8):
            x = x + sl.cho_solve((L, True), (b - Kl @ x).astype(np.float64)).astype(np.longdouble)
        return x
    T = np.stack([solve_ld(kl[:, i]) for i in range(Kuf.shape[1])], 1)
    var_t = np.longdouble(var) - np.sum(kl * T, 0) + np.sum((s[:, None].astype(np.longdouble) ** 2) * T * T, 0)
    A1 = sl.solve_triangular(L, Kuf, lower=True)
    A2 = sl.solve_triangular(L.T, A1, lower=False)
    var_w = var - np.sum(A1 ** 2, 0) + np.sum((s[:, None] * A2) ** 2, 0)
    eb = float(np.max(np.abs(var_b - var_t) / np.abs(var_t)))
    ew = float(np.max(np.abs(var_w - var_t) / np.abs(var_t)))
    print('%-26s M %4d cond(Kuu) %.2e | relative error of var: B-form %.2e, W-form %.2e' % (name, M, cond, eb, ew))


if __name__ == '__main__':
    X, Y, p = bench.synth(4096, 1024, 3)
    study('cfg3 f', p['Zf'], p['ell_f'], 1.0, np.ones(1024), X)
    study('cfg3 g', p['Zg'], p['ell_g'], 5.0, np.ones(1024), X)
    X, Y, p = bench.synth(4096, 512, 3)
    study('cfg2 f', p['Zf'], p['ell_f'], 1.0, np.ones(512), X)
    study('cfg2 g', p['Zg'], p['ell_g'], 5.0, np.ones(512), X)
    for (N, M, D, ell) in ((2048, 128, 3, 0.3), (3000, 200, 3, 0.25), (1500, 300, 2, 0.2), (1500, 96, 4, 0.5), (1300, 150, 8, 0.9)):
        X, Y, p = make_problem(N, M, D, seed=N + M, ell=ell)
        study('tests %d/%d/D%d f' % (N, M, D), p['Zf'], p['ell_f'], p['var_f'], p['u_fs_sqrt'].reshape(-1), X)
        study('tests %d/%d/D%d g' % (N, M, D), p['Zg'], p['ell_g'], p['var_g'], p['u_gs_sqrt'].reshape(-1), X)
