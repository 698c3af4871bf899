"""GPU parity of the Kronecker (space x time) path against the LITERAL dense restatement of
scripts/onoff.py:143-319 / onofftf/main.py:350-387 (oracle/zigp_oracle*.py)."""
import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu


def make_kron_problem(N, M0, M1, seed=0, M0g=None, M1g=None, ell_s=3.0, ell_t=None, u_scale=0.1):
    """pptr-like: 2 spatial columns, 1 temporal column; ~60 % exact zeros in Y."""
    # temporal lengthscale ~ 1.5x the inducing spacing, as in the reference's init (linspace grid, scripts/onoff.py:57,68)
    ell_t = ell_t if ell_t is not None else 1.5 / max(M1 - 1, 1)
    rs = np.random.RandomState(seed)
    X = np.hstack([rs.rand(N, 2) * 10.0, rs.rand(N, 1)])
    f = np.sin(X[:, 0]) + np.cos(3 * X[:, 2])
    Y = np.where(rs.rand(N) > 0.6, np.abs(f + 0.3 * rs.randn(N)), 0.0)[:, None]
    M0g, M1g = M0g or M0, M1g or M1
    p = dict(Zf=[rs.rand(M0, 2) * 10.0, np.linspace(0, 1, M1)[:, None]], Zg=[rs.rand(M0g, 2) * 10.0, np.linspace(0, 1, M1g)[:, None]],
             ell_f=[np.array([ell_s, ell_s * 1.2]), np.array([ell_t])], ell_g=[np.array([ell_s * 0.8, ell_s]), np.array([ell_t * 1.5])],
             var_f=[np.array([2.0]), np.array([1.5])], var_g=[np.array([1.2]), np.array([0.9])],
             u_fm=u_scale * rs.randn(M0 * M1, 1), u_gm=u_scale * rs.randn(M0g * M1g, 1),
             u_fs_sqrt=0.5 + rs.rand(M0 * M1, 1), u_gs_sqrt=0.5 + rs.rand(M0g * M1g, 1), noise=0.05)
    return X, Y, p


ELL_T_HARD = 0.3   # 32 inducing times on [0,1] with this lengthscale: cond(K_t) ~ 1e7 at jitter 1e-5
CASES = [(200, 6, 5, None, None), (1000, 10, 12, 8, 9), (700, 32, 32, None, None), (1500, 10, 100, None, None)]


def _mp_kron_inf(X, Zl, ell, var, u, s, jitter, npts):
    """40-digit evaluation of kron_inf (scripts/onoff.py:186-213) for the first npts rows: 'truth' when the factor
    matrices are so ill-conditioned that the float64 oracle (LU inverse, :192) and the GPU (Cholesky) both drift."""
    import mpmath as mp
    mp.mp.dps = 40

    def kmat(A, B, l, v):
        return mp.matrix([[mp.mpf(float(v)) * mp.exp(-sum(((mp.mpf(float(a[d])) - mp.mpf(float(b[d]))) / mp.mpf(float(l[d]))) ** 2
                                                           for d in range(len(l))) / 2) for b in B] for a in A])
    P, ks = [], []
    c0 = 0
    for q in range(2):
        Z = Zl[q]
        K = kmat(Z, Z, ell[q], var[q])
        for i in range(Z.shape[0]):
            K[i, i] += mp.mpf(jitter)
        P.append(K ** -1)
        ks.append(kmat(Z, X[:npts, c0:c0 + Z.shape[1]], ell[q], var[q]))
        c0 += Z.shape[1]
    M0, M1 = Zl[0].shape[0], Zl[1].shape[0]
    U = mp.matrix(M0, M1)
    S2 = mp.matrix(M0, M1)
    for i in range(M0):
        for j in range(M1):
            U[i, j] = mp.mpf(float(u[i * M1 + j, 0]))
            S2[i, j] = mp.mpf(float(s[i * M1 + j, 0])) ** 2
    Al = P[0] * U * P[1]
    a0, a1 = P[0] * ks[0], P[1] * ks[1]
    mu, vv = [], []
    knn = mp.mpf(float(var[0])) * mp.mpf(float(var[1]))
    for n in range(npts):
        k0, k1 = ks[0][:, n], ks[1][:, n]
        m = (k0.T * Al * k1)[0]
        q0 = sum(k0[i] * a0[i, n] for i in range(M0))
        q1 = sum(k1[j] * a1[j, n] for j in range(M1))
        t0 = mp.matrix([a0[i, n] ** 2 for i in range(M0)])
        t1 = mp.matrix([a1[j, n] ** 2 for j in range(M1)])
        st = (t0.T * S2 * t1)[0]
        mu.append(float(m))
        vv.append(float(knn - q0 * q1 + st))
    return np.array(mu), np.array(vv)


def _factored_with_oracle_inverse(X, p, tag, jit):
    """The FACTORED identities the engine evaluates, on the CPU with the oracle's own np.linalg.inv (LAPACK LU, as
    tf.matrix_inverse scripts/onoff.py:192): its distance from the literal dense order is the floor that the op order alone
    sets (tests/test_cpu_oracle.py::test_factored_kronecker_algebra_differs_...; tools/lu_vs_chol_experiment.py)."""
    import zigp_oracle as o
    Z, ell, var = p['Z' + tag], p['ell_' + tag], [float(np.squeeze(v)) for v in p['var_' + tag]]
    P = [np.linalg.inv(o.rbf_K(Z[q], None, ell[q], var[q]) + jit * np.eye(Z[q].shape[0])) for q in range(2)]
    d0 = Z[0].shape[1]
    k0, k1 = o.rbf_K(Z[0], X[:, :d0], ell[0], var[0]), o.rbf_K(Z[1], X[:, d0:], ell[1], var[1])
    M0, M1 = Z[0].shape[0], Z[1].shape[0]
    U, S2 = p['u_%sm' % tag].reshape(M0, M1), np.square(p['u_%ss_sqrt' % tag]).reshape(M0, M1)
    a0, a1 = P[0] @ k0, P[1] @ k1
    mu = np.einsum('in,ij,jn->n', k0, P[0] @ U @ P[1], k1)
    vv = var[0] * var[1] - (k0 * a0).sum(0) * (k1 * a1).sum(0) + np.einsum('in,ij,jn->n', a0 ** 2, S2, a1 ** 2)
    return mu, vv


@pytest.mark.parametrize('N,M0,M1,M0g,M1g,HARD', [c + (False,) for c in CASES] + [(700, 32, 32, None, None, True)])
def test_kron_predict_matches_literal_oracle(engine, N, M0, M1, M0g, M1g, HARD):
    """Well-conditioned factors: GPU == literal oracle to 1e-6 (north-star tolerance, fp64).  Ill-conditioned ones (32 / 100 points on
    a line with a long lengthscale, cond(K_p) ~ 1e6-1e7): the factored op order itself -- even with the oracle's own LU inverse --
    sits a few 1e-6 from the literal dense order, so there the GPU is held to (i) that op-order floor: within 3x of the distance of
    the CPU factored evaluation from the oracle, and (ii) a 40-digit evaluation: not worse than 10x the oracle's own error."""
    import zigp_oracle as o
    X, Y, p = make_kron_problem(N, M0, M1, seed=N, M0g=M0g, M1g=M1g, ell_t=ELL_T_HARD if HARD else None)
    hard = HARD
    for jit, goff in (((1e-5, 0.0),) if hard else ((1e-5, 0.0), (1e-6, -1.0))):
        out = engine.kron_predict(p, X, jitter=jit, g_offset=goff)
        ref = o.kron_build_predict(X, p, jit, goff)
        names = ('gfmean', 'gfvar', 'gfmeanu', 'fmean', 'fvar', 'gmean', 'gvar', 'ephi_g', 'evar_phi_g')
        errs = [relerr(out[i], ref[i].reshape(-1)) for i in range(9)]
        for name, e in zip(names, errs):
            print('jitter %g %s relerr %.2e' % (jit, name, e))
        if not hard and max(errs) < 1e-6:
            continue
        npts = 12
        for tag, (im, iv) in (('f', (3, 4)), ('g', (5, 6))):
            cm, cv = _factored_with_oracle_inverse(X, p, tag, jit)
            if tag == 'g':
                cm = cm + goff
            for nm, idx, cpu in (('mean', im, cm), ('var', iv, cv)):
                floor = relerr(cpu, ref[idx].reshape(-1))
                e = relerr(out[idx], ref[idx].reshape(-1))
                print('jitter %g %s%s: gpu vs oracle %.2e, op-order floor (CPU factored, np.linalg.inv) %.2e' % (jit, tag, nm, e, floor))
                assert e < max(1e-6, 3.0 * floor), (tag, nm, e, floor)
            tm, tv = _mp_kron_inf(X, p['Z' + tag], p['ell_' + tag], [float(np.squeeze(v)) for v in p['var_' + tag]],
                                  p['u_%sm' % tag], p['u_%ss_sqrt' % tag], jit, npts)
            if tag == 'g':
                tm = tm + goff
            for nm, idx, truth in (('mean', im, tm), ('var', iv, tv)):
                e_gpu = relerr(out[idx][:npts], truth)
                e_orc = relerr(ref[idx].reshape(-1)[:npts], truth)
                print('jitter %g %s%s: gpu vs 40-digit %.2e, oracle vs 40-digit %.2e' % (jit, tag, nm, e_gpu, e_orc))
                assert e_gpu < max(1e-6, 10 * e_orc), (tag, nm, e_gpu, e_orc)


@pytest.mark.parametrize('N,M0,M1,M0g,M1g', CASES[:3] + [(600, 10, 100, None, None), (500, 16, 112, 9, 50)])
def test_kron_elbo_and_gradient_match_literal_oracle(engine, N, M0, M1, M0g, M1g):
    import zigp_oracle_torch as ot
    X, Y, p = make_kron_problem(N, M0, M1, seed=N + 1, M0g=M0g, M1g=M1g)
    scale = 105280.0 / N
    ed, kl, g = engine.kron_elbo(p, X, Y, jitter=1e-5, scale=scale)
    e_r, d_r, kl_r, g_r = ot.kron_elbo_and_grad(X, Y, p, 1e-5, scale=scale)
    print('elbo %.10e ref %.10e  kl %.8e ref %.8e' % (ed - kl, e_r, kl, kl_r))
    assert abs(ed - scale * d_r) <= 1e-7 * abs(scale * d_r)
    assert abs(kl - kl_r) <= 1e-7 * abs(kl_r)
    for k in ('Zf', 'Zg', 'ell_f', 'ell_g', 'var_f', 'var_g'):
        for q in range(2):
            a, b = np.asarray(g[k][q]).reshape(-1), np.asarray(g_r[k][q]).reshape(-1)
            e = np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)
            print('  grad %s[%d] relerr %.2e (max |ref| %.3e)' % (k, q, e, np.max(np.abs(b))))
            assert e < 1e-6, (k, q, e)
    for k in ('u_fm', 'u_gm', 'u_fs_sqrt', 'u_gs_sqrt', 'noise'):
        a, b = np.asarray(g[k]).reshape(-1), np.asarray(g_r[k]).reshape(-1)
        e = np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)
        print('  grad %s relerr %.2e' % (k, e))
        assert e < 1e-6, (k, e)


def test_kron_value_only_and_no_kl(engine):
    X, Y, p = make_kron_problem(300, 5, 4, seed=3)
    ed, kl, g = engine.kron_elbo(p, X, Y, need_grad=False)
    assert g is None and np.isfinite(ed) and kl > 0
    ed2, kl2, _ = engine.kron_elbo(p, X, Y, include_kl=False, need_grad=False)
    assert kl2 == 0.0 and ed2 == ed


def test_kron_elbo_on_resident_rows_equals_host_minibatch(engine):
    """zigp_kron_elbo_rows: rows of the resident data set instead of a host minibatch -- same kernels, same numbers"""
    X, Y, p = make_kron_problem(3000, 32, 32, seed=5)
    engine.set_data(X, Y)
    for lo, hi in ((0, 3000), (1000, 2000), (17, 1234)):
        a = engine.kron_elbo(p, X[lo:hi], Y[lo:hi], jitter=1e-5, scale=2.5)
        b = engine.kron_elbo(p, rows=(lo, hi), jitter=1e-5, scale=2.5)
        assert a[0] == b[0] and a[1] == b[1]
        for k in a[2]:
            va, vb = a[2][k], b[2][k]
            if isinstance(va, list):
                for x, y in zip(va, vb):
                    assert np.array_equal(np.asarray(x), np.asarray(y)), k
            else:
                assert np.array_equal(np.asarray(va), np.asarray(vb)), k
    with pytest.raises(ValueError):
        engine.kron_elbo(p, rows=(7, 5))          # (an EMPTY range is legal since round 4: test_kron_empty_row_range_...)
    with pytest.raises(ValueError):
        engine.kron_elbo(p, rows=(0, 3001))


@pytest.mark.parametrize('N,M0,M1,M0g,M1g', [(900, 32, 32, None, None), (1200, 10, 100, None, None), (400, 6, 5, 7, 4), (700, 16, 112, 9, 50)])
def test_kron_fused_and_panel_paths_agree(engine, N, M0, M1, M0g, M1g):
    """Two independent implementations of the same factored algebra -- the fused register-resident MFMA kernels (zigp_kronf.hip: small
    grids in registers, the reference's 10 x 100 through spill + accumulate) and the GEMM-panel path (zigp_kron.hip) -- on identical
    inputs: value, KL, every gradient block and the 9-tuple of predictions.  Also: reruns are bit-identical (fixed-order reductions)."""
    X, Y, p = make_kron_problem(N, M0, M1, seed=9 + M1, M0g=M0g, M1g=M1g)
    a = engine.kron_elbo(p, X, Y, jitter=1e-5, scale=3.0)
    a2 = engine.kron_elbo(p, X, Y, jitter=1e-5, scale=3.0)
    pa = engine.kron_predict(p, X, jitter=1e-6, g_offset=-1.0)
    engine.set_kron_panels(True)
    try:
        b = engine.kron_elbo(p, X, Y, jitter=1e-5, scale=3.0)
        pb = engine.kron_predict(p, X, jitter=1e-6, g_offset=-1.0)
    finally:
        engine.set_kron_panels(False)
    assert a[0] == a2[0] and a[1] == a2[1] and np.array_equal(a[2]['u_fm'], a2[2]['u_fm']) and np.array_equal(a[2]['Zg'][1], a2[2]['Zg'][1])
    assert abs(a[0] - b[0]) <= 1e-8 * abs(b[0]) and abs(a[1] - b[1]) <= 1e-9 * abs(b[1])   # two op orders, cond(K_p) up to 1e6
    for k in ('u_fm', 'u_gm', 'u_fs_sqrt', 'u_gs_sqrt', 'noise'):
        assert relerr(a[2][k], b[2][k]) < 1e-7, k
    for k in ('Zf', 'Zg', 'ell_f', 'ell_g', 'var_f', 'var_g'):
        for q in range(2):
            assert relerr(a[2][k][q], b[2][k][q]) < 1e-7, (k, q)
    for i in range(9):
        assert relerr(pa[i], pb[i]) < 1e-6, i      # prediction jitter 1e-6: cond(K_p) * eps through two op orders (north-star tolerance)


@pytest.mark.parametrize('M0,M1', [(32, 32), (10, 100)])
def test_kron_shard_additivity_many_tiles(engine, M0, M1):
    """20 000 rows (1 250 tiles: several tiles per wave, many partial accumulators / splits): the data term and its gradient are
    sums over points, so two halves (KL once) must add up to the full batch -- exercises the tile / split bookkeeping of both
    kernel variants (registers for 32 x 32, spill + accumulate for the reference's 10 x 100)."""
    X, Y, p = make_kron_problem(20000, M0, M1, seed=11)
    ed, kl, g = engine.kron_elbo(p, X, Y, jitter=1e-5, scale=1.0)
    a = engine.kron_elbo(p, X[:9000], Y[:9000], jitter=1e-5, scale=1.0, include_kl=True)
    b = engine.kron_elbo(p, X[9000:], Y[9000:], jitter=1e-5, scale=1.0, include_kl=False)
    assert abs((a[0] + b[0]) - ed) <= 1e-11 * abs(ed) and a[1] == kl and b[1] == 0.0
    for k in ('u_fm', 'u_gm', 'u_fs_sqrt', 'u_gs_sqrt', 'noise'):
        s_ = np.asarray(a[2][k]) + np.asarray(b[2][k])
        assert np.max(np.abs(s_ - np.asarray(g[k]))) <= 1e-9 * max(np.max(np.abs(np.asarray(g[k]))), 1e-300), k
    for k in ('Zf', 'Zg', 'ell_f', 'ell_g', 'var_f', 'var_g'):
        for q in range(2):
            s_ = np.asarray(a[2][k][q]) + np.asarray(b[2][k][q])
            ref = np.asarray(g[k][q])
            assert np.max(np.abs(s_ - ref)) <= 1e-8 * max(np.max(np.abs(ref)), 1e-300), (k, q)
    # and a 3000-row slice against the literal oracle is covered by the parametrised tests above


def test_kron_larger_grid_row_ranges_are_bit_stable(engine):
    """The gradient step on the reference's [10, 100] grid (scripts/onoff.py:52-53) sends its rows through in ranges (bounded operand
    spill, zigp_kronf.hip): 20 000 rows = 1 250 tiles go through as 2 ranges by default and as 20+ with 57-tile ranges -- every result
    must be the same bits, and a row count that used to need 0.9 GB of records (the pptr full batch, 105 280 rows) runs in 128 MB."""
    X, Y, p = make_kron_problem(20000, 10, 100, seed=12)
    ref = engine.kron_elbo(p, X, Y, jitter=1e-5, scale=2.0)
    try:
        for tiles in (57, 64, 100000):
            engine.set_kron_range_tiles(tiles)
            got = engine.kron_elbo(p, X, Y, jitter=1e-5, scale=2.0)
            assert got[0] == ref[0] and got[1] == ref[1], tiles
            for k in ref[2]:
                a, b = got[2][k], ref[2][k]
                for x, y in (zip(a, b) if isinstance(b, (list, tuple)) else ((a, b),)):
                    assert np.array_equal(np.asarray(x), np.asarray(y)), (tiles, k)
    finally:
        engine.set_kron_range_tiles(1024)
    Xb, Yb, pb = make_kron_problem(105280, 10, 100, seed=13)          # pptr's row count
    ed, kl, g = engine.kron_elbo(pb, Xb, Yb, jitter=1e-5, scale=1.0)
    h = 52640
    a = engine.kron_elbo(pb, Xb[:h], Yb[:h], jitter=1e-5, scale=1.0, include_kl=True)
    b = engine.kron_elbo(pb, Xb[h:], Yb[h:], jitter=1e-5, scale=1.0, include_kl=False)
    assert abs((a[0] + b[0]) - ed) <= 1e-11 * abs(ed) and a[1] == kl
    for q in range(2):
        s_ = np.asarray(a[2]['Zf'][q]) + np.asarray(b[2]['Zf'][q])
        assert np.max(np.abs(s_ - np.asarray(g['Zf'][q]))) <= 1e-8 * np.max(np.abs(np.asarray(g['Zf'][q])))


@pytest.mark.parametrize('M0,M1', [(32, 32), (10, 100), (40, 9)])
def test_kron_empty_row_range_contributes_zero_and_keeps_the_kl(engine, M0, M1):
    """A rank of a data-parallel run whose shard is empty still makes the same calls (the exchange inside the step is collective):
    rows=(k, k) gives a zero data term, the KL as usual, and exactly the KL part of the gradient -- fused small grid, larger grid, panels."""
    X, Y, p = make_kron_problem(600, M0, M1, seed=14)
    engine.set_data(X, Y)
    full = engine.kron_elbo(p, rows=(0, 600), jitter=1e-5, scale=3.0)
    nokl = engine.kron_elbo(p, rows=(0, 600), jitter=1e-5, scale=3.0, include_kl=False)
    for k0 in (0, 17, 600):
        e = engine.kron_elbo(p, rows=(k0, k0), jitter=1e-5, scale=3.0)
        assert e[0] == 0.0 and e[1] == full[1]
        z = engine.kron_elbo(p, rows=(k0, k0), jitter=1e-5, scale=3.0, include_kl=False)
        assert z[0] == 0.0 and z[1] == 0.0
        for k in full[2]:
            a, b, c, d = e[2][k], full[2][k], nokl[2][k], z[2][k]
            items = zip(a, b, c, d) if isinstance(b, (list, tuple)) else ((a, b, c, d),)
            for x, y, w, v in items:
                x, y, w, v = (np.asarray(t, dtype=float).reshape(-1) for t in (x, y, w, v))
                assert np.all(v == 0.0), k
                assert np.max(np.abs(x - (y - w))) <= 1e-9 * max(np.max(np.abs(y)), 1e-300), k      # KL part = full - data part


def test_kron_predict_chunks_rows(engine):
    """prediction sets larger than 131072 rows are processed in chunks (bounded device / pinned memory): same values as per-chunk calls"""
    X, Y, p = make_kron_problem(140000, 12, 9, seed=2)
    out = engine.kron_predict(p, X, jitter=1e-6, g_offset=-1.0)
    a = engine.kron_predict(p, X[:131072], jitter=1e-6, g_offset=-1.0)
    b = engine.kron_predict(p, X[131072:], jitter=1e-6, g_offset=-1.0)
    assert out.shape == (9, 140000) and np.array_equal(out[:, :131072], a) and np.array_equal(out[:, 131072:], b)
    ph = {k: p[k] for k in ('Zf', 'ell_f', 'var_f', 'u_fm', 'u_fs_sqrt', 'noise')}
    o4 = engine.kron_head_predict(ph, X, 'gaussian', jitter=1e-6)
    assert o4.shape == (4, 140000) and np.array_equal(o4[:, 131072:], engine.kron_head_predict(ph, X[131072:], 'gaussian', jitter=1e-6))


@pytest.mark.parametrize('M0,M1', [(8, 6), (10, 100), (40, 9)])
def test_kron_not_positive_definite_factor_raises(engine, M0, M1):
    """tf.cholesky raises InvalidArgumentError on a non-PD factor (onofftf/main.py:355); the engine returns ZIGP_ENOTPD on every
    path (fused small grid, larger grid, GEMM panels) and keeps working afterwards.  Duplicate inducing points with jitter 0."""
    import zigp
    X, Y, p = make_kron_problem(300, M0, M1, seed=4)
    bad = dict(p, Zf=[p['Zf'][0].copy(), p['Zf'][1].copy()])
    bad['Zf'][1][-1] = bad['Zf'][1][0]
    with pytest.raises(zigp.NotPositiveDefiniteError):
        engine.kron_elbo(bad, X, Y, jitter=0.0)
    with pytest.raises(zigp.NotPositiveDefiniteError):
        engine.kron_predict(bad, X, jitter=0.0)
    ed, kl, g = engine.kron_elbo(p, X, Y, jitter=1e-5)
    assert np.isfinite(ed) and np.isfinite(kl)


def test_kron_tiny_and_ragged_batches(engine):
    """fewer rows than one 16-point tile, a ragged tile, one inducing point per factor"""
    import zigp_oracle as o
    for N, M0, M1 in ((1, 4, 3), (5, 1, 1), (17, 2, 33), (1000, 32, 1)):
        X, Y, p = make_kron_problem(N, M0, M1, seed=N)
        out = engine.kron_predict(p, X, jitter=1e-5)
        ref = o.kron_build_predict(X, p, 1e-5, 0.0)
        for i in range(9):
            assert relerr(out[i], ref[i].reshape(-1)) < 1e-7, (N, M0, M1, i)
        ed, kl, g = engine.kron_elbo(p, X, Y, jitter=1e-5)
        e_r, d_r, klf, klg = o.kron_elbo(X, Y, p, 1e-5)
        assert abs(ed - d_r) <= 1e-8 * max(abs(d_r), 1.0) and abs(kl - (klf + klg)) <= 1e-8 * abs(klf + klg)


@pytest.mark.parametrize('M0,M1', [(32, 32), (10, 100)])
def test_kron_stepper_equals_unprepared_call(engine, M0, M1):
    """DenseEngine.kron_stepper: the fit loop's prepared step (buffers and structs set up once for a fixed model shape) goes through the
    same entry points as kron_elbo -- same numbers bit for bit, for changing parameter values, host minibatches and resident rows, f_mu."""
    X, Y, p = make_kron_problem(2500, M0, M1, seed=21)
    st = engine.kron_stepper(p)
    engine.set_data(X, Y)
    rs = np.random.RandomState(0)
    for it in range(3):
        q = dict(p, u_fm=p['u_fm'] + 0.01 * it * rs.randn(*p['u_fm'].shape), noise=0.05 + 0.01 * it,
                 var_g=[p['var_g'][0] * (1 + 0.1 * it), p['var_g'][1]], ell_f=[p['ell_f'][0] * (1 + 0.05 * it), p['ell_f'][1]])
        for kw in (dict(X=X[:1000], Y=Y[:1000]), dict(rows=(300, 2400)), dict(X=X[:777], Y=Y[:777], f_mu=0.2)):
            a = engine.kron_elbo(q, jitter=1e-5, scale=3.0, **kw)
            b = st(q, jitter=1e-5, scale=3.0, **kw)
            assert a[0] == b[0] and a[1] == b[1] and set(a[2]) == set(b[2]), (it, list(kw))
            for k in a[2]:
                va, vb = a[2][k], b[2][k]
                for x, y in (zip(va, vb) if isinstance(va, list) else ((va, vb),)):
                    assert np.array_equal(np.asarray(x), np.asarray(y)), (it, k)
    with pytest.raises(ValueError):
        st(dict(p, u_fm=np.zeros(3)), X[:10], Y[:10])       # another model shape needs another stepper
