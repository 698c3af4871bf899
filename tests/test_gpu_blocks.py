"""GPU parity of the building blocks (through the C-ABI diagnostics): fp64 MFMA GEMM core in every
operand layout, blocked Cholesky + triangular inverse, RBF kernel matrix."""
import numpy as np
import pytest
import scipy.linalg as sl

from conftest import relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('ta,tb', [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize('m,n,k', [(128, 128, 16), (200, 300, 70), (384, 256, 512), (1, 1, 1)])
def test_gemm_layouts(engine, ta, tb, m, n, k):
    rs = np.random.RandomState(m + 3 * n + 7 * k)
    A = rs.randn(k, m) if ta else rs.randn(m, k)
    B = rs.randn(n, k) if tb else rs.randn(k, n)
    ref = (A.T if ta else A) @ (B.T if tb else B)
    C = engine.test_gemm(A, B, ta, tb)
    # asymmetric integer check as well (exact): catches row/col swaps
    assert relerr(C, ref) < 1e-13
    Ai = np.arange(A.size, dtype=np.float64).reshape(A.shape) % 7 - 3
    Bi = (np.arange(B.size, dtype=np.float64).reshape(B.shape) * 3) % 5 - 2
    Ci = engine.test_gemm(Ai, Bi, ta, tb)
    assert np.array_equal(Ci, (Ai.T if ta else Ai) @ (Bi.T if tb else Bi))


@pytest.mark.parametrize('split_k', [False, True])   # one workgroup per tile (Kronecker panel path) / k slices + ordered reduction (dense M x M forward)
@pytest.mark.parametrize('n', [1, 9, 31, 32, 33, 50, 64, 65, 96, 97, 100, 127, 128, 129, 200, 512, 640, 1024])   # every panel count and boundary of the blocked factorisation
def test_potrf_trtri(engine, n, split_k):
    rs = np.random.RandomState(n)
    Z = rs.rand(n, 3)
    d2 = ((Z[:, None, :] - Z[None, :, :]) ** 2).sum(-1)
    A = np.exp(-0.5 * d2 / 0.1 ** 2) + 1e-6 * np.eye(n)
    L, W = engine.test_potrf_trtri(A, split_k)
    Lr = sl.cholesky(A, lower=True)
    assert np.all(np.triu(L, 1) == 0) and np.all(np.triu(W, 1) == 0)
    assert relerr(L @ L.T, A) < 1e-13
    assert relerr(L, Lr) < 1e-9          # cond(A) amplifies rounding differences between summation orders
    # W is the inverse of L to backward-stable accuracy
    assert np.max(np.abs(W @ L - np.eye(n))) < 1e-9 * max(1.0, np.linalg.cond(L) * 1e-6)


def test_potrf_not_pd(engine):
    import zigp
    A = np.eye(200)
    A[150, 150] = -1.0
    with pytest.raises(zigp.NotPositiveDefiniteError):
        engine.test_potrf_trtri(A)
    # the context stays usable afterwards
    L, _ = engine.test_potrf_trtri(np.eye(5) * 4.0)
    assert np.allclose(L, 2 * np.eye(5))


@pytest.mark.parametrize('D,ard', [(1, False), (3, False), (3, True)])
def test_rbf_K_matches_oracle(engine, D, ard):
    import zigp_oracle as o
    rs = np.random.RandomState(D)
    X1, X2 = rs.rand(77, D) * 3, rs.rand(130, D) * 3
    ell = (0.5 + rs.rand(D)) if ard else np.array([0.7])
    var = 2.5
    K = engine.rbf_K(X1, X2, ell, var)
    assert relerr(K, o.rbf_K(X1, X2, ell, var)) < 1e-12
    Ks = engine.rbf_K(X1, None, ell, var)
    assert relerr(Ks, o.rbf_K(X1, None, ell, var)) < 1e-12


def test_kuf_panel_kernel_golden_kernse_np_and_ulp(engine):
    """The chunk loop's Kuf kernel (k_kuf_build: exponential written out by hand -- 32-entry table, degree-6 polynomial, ldexp) against
    the reference's own kernse_np outputs (G1), and against an 80-bit evaluation over 800 units of -log K: within a few ulp plus the
    conditioning of exp (the argument itself carries a few roundings: relative error ~ eps * y), gradual underflow, exact zero beyond."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g1_kernse_np.npz'))
    for tag in ('d1', 'd3s', 'd3ard'):
        Z, X, ell, var = g[tag + '_Z'], g[tag + '_X'], g[tag + '_ell'], float(g[tag + '_var'])
        assert relerr(engine.test_kuf(X, Z, ell, var), g[tag + '_Kzx']) < 1e-12
    rng = np.random.RandomState(11)
    eps = np.finfo(np.float64).eps
    for D in (1, 2, 3, 5, 8):
        N, M = 3000 + 37 * D, 70 + D
        X = rng.randn(N, D) * rng.choice([0.3, 3.0, 30.0], size=(N, 1))
        Z = rng.randn(M, D)
        ell = 0.5 + rng.rand(D)
        var = 1.7
        K = engine.test_kuf(X, Z, ell, var)
        L = np.longdouble
        d = (Z.astype(L)[:, None, :] - X.astype(L)[None, :, :]) / ell.astype(L)
        y = L(0.5) * (d * d).sum(-1)
        ref = L(var) * np.exp(-y)
        tol = (4.0 + 8.0 * y) * eps * ref + L(5e-324)
        bad = np.abs(K.astype(L) - ref) > tol
        assert not bad.any(), (D, int(bad.sum()), float(np.max(np.abs(K.astype(L) - ref) / (ref + L(1e-300)))))
        assert (K[np.asarray(y > 760)] == 0.0).all() and (K >= 0).all() and np.isfinite(K).all()
        assert float(y.max()) > 800 or D < 3        # the sweep reaches past the underflow threshold


def test_rbf_K_golden_kernse_np(engine):
    """Golden vectors produced by the reference's own kernse_np (onofftf/utils.py:26-58)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g1_kernse_np.npz'))
    for tag in ('d1', 'd3s', 'd3ard'):
        Z, X, ell, var = g[tag + '_Z'], g[tag + '_X'], g[tag + '_ell'], float(g[tag + '_var'])
        assert relerr(engine.rbf_K(Z, X, ell, var), g[tag + '_Kzx']) < 1e-12
        assert relerr(engine.rbf_K(Z, None, ell, var), g[tag + '_Kzz']) < 1e-12
