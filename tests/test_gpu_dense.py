"""GPU parity of the dense ELBO path (predict 9-tuple, ELBO, KL, full gradient) against the CPU oracle.

Tolerances: north_star asks for 1e-6 relative (fp64) on ELBO and predictive mean/var; the tests use
tighter bounds where cond(Kuu) allows and print cond(Kuu) next to each case."""
import numpy as np
import pytest

from conftest import make_problem, relerr

pytestmark = pytest.mark.gpu

CASES = [
    # N, M, Mg, D, ell, chunk, u_scale
    (450, 9, 9, 1, 2.0, None, 0.5),        # cfg1-like (notebook: 9 inducing points)
    (450, 50, 50, 1, 2.0, None, 0.01),     # cfg1 of BASELINE.json (M=50): cond(Kuu)~2e7, u_m at the reference's init scale (OnOffSVGP.py:56-57)
    (2048, 128, 128, 3, 0.3, None, 0.5),   # cut-down cfg2
    (3000, 200, 136, 3, 0.25, 1024, 0.5),  # ragged: M not a multiple of 128, Mf != Mg, 3 chunks with a partial last one
    (1500, 300, 300, 2, 0.2, 1024, 0.5),
    (1200, 300, 100, 2, 0.25, 1024, 0.5),  # three 128-blocks against one: the two latents' factorisation chains (launched alternately) differ in length
    (1200, 100, 520, 2, 0.25, 1024, 0.5),  # one block against five (the longer chain on the second stream)
    # every instantiated input dimension (k_kuf_build<D>, k_kgrad<D>, D = 1..8; ARD lengthscales: the broadcast divide of onofftf/main.py:42)
    (1500, 96, 140, 4, 0.5, 1024, 0.5),
    (1400, 150, 90, 5, 0.6, 1024, 0.5),
    (1100, 130, 64, 6, 0.7, 1024, 0.5),
    (1000, 64, 64, 7, 0.8, None, 0.5),
    (1300, 150, 100, 8, 0.9, 1024, 0.5),
]


def _cond(p, jitter):
    import zigp_oracle as o
    K = o.rbf_K(p['Zf'], None, p['ell_f'], p['var_f']) + jitter * np.eye(p['Zf'].shape[0])
    return np.linalg.cond(K)


@pytest.mark.parametrize('N,M,Mg,D,ell,chunk,us', CASES)
def test_predict_matches_oracle(engine, N, M, Mg, D, ell, chunk, us):
    import zigp_oracle as o
    X, Y, p = make_problem(N, M, D, seed=N + M, Mg=Mg, ell=ell, u_scale=us)
    if D == 1:
        X = X * 10.0
        p['Zf'] *= 10.0
        p['Zg'] *= 10.0
    engine.set_chunk(chunk or 16384)
    for g_off in (0.0, -1.0):
        out = engine.predict(p, X, jitter=1e-6, g_offset=g_off)
        ref = o.build_predict(X, p, 1e-6, g_off)
        c = _cond(p, 1e-6)
        tol = max(1e-9, 1e-13 * c)
        for i, name in enumerate(('gfmean', 'gfvar', 'gfmeanu', 'fmean', 'fvar', 'gmean', 'gvar', 'ephi_g', 'evar_phi_g')):
            e = relerr(out[i], ref[i].reshape(-1))
            print('cond(Kuu)=%.2e %s relerr=%.2e' % (c, name, e))
            assert e < min(tol, 1e-6), (name, e, c)


def test_predict_device_matches_oracle_and_the_host_entry_point(engine):
    """zigp_predict_device (device X in, device (9,N) out: the companion of zigp_set_data_device) against the oracle's build_predict
    (onoffgpf/OnOffSVGP.py:124-152) and bit for bit against zigp_predict; a host pointer is rejected, not dereferenced on the device."""
    import ctypes as C
    import torch
    import zigp_oracle as o
    from zigp import _lib
    from zigp.engine import _Packed
    X, Y, p = make_problem(3001, 70, 3, seed=12, Mg=45, ell=0.4)
    engine.set_chunk(1024)                                  # three passes, the last one partial
    Xd = torch.from_numpy(X).to('cuda:0')
    for g_off in (0.0, -1.0):
        out_d = engine.predict_device(p, Xd, jitter=1e-6, g_offset=g_off)
        assert out_d.is_cuda and tuple(out_d.shape) == (9, 3001)
        out_h = engine.predict(p, X, jitter=1e-6, g_offset=g_off)
        assert np.array_equal(out_d.cpu().numpy(), out_h)
        ref = o.build_predict(X, p, 1e-6, g_off)
        tol = max(1e-9, 1e-13 * _cond(p, 1e-6))
        for i in range(9):
            assert relerr(out_h[i], ref[i].reshape(-1)) < min(tol, 1e-6), i
    buf = torch.full((9, 3001), -7.0, dtype=torch.float64, device='cuda:0')
    assert engine.predict_device(p, Xd, out=buf) is buf and np.array_equal(buf.cpu().numpy(), engine.predict(p, X))
    assert tuple(engine.predict_device(p, Xd[:0]).shape) == (9, 0)
    with pytest.raises(ValueError):
        engine.predict_device(p, Xd.cpu())
    with pytest.raises(ValueError):
        engine.predict_device(p, Xd.float())
    pk = _Packed(p)
    host = np.zeros((9, 8))
    rc = engine.lib.zigp_predict_device(engine.ctx, C.byref(pk.struct), C.c_void_p(X.ctypes.data), 8, 1e-6, 0.0, C.c_void_p(host.ctypes.data))
    assert rc == _lib.ZIGP_EARG and b'device memory' in engine.lib.zigp_last_error(engine.ctx)
    engine.set_chunk(16384)


@pytest.mark.parametrize('N,M,Mg,D,ell,chunk,us', CASES)
def test_elbo_and_gradient_match_oracle(engine, N, M, Mg, D, ell, chunk, us):
    import zigp_oracle_torch as ot
    X, Y, p = make_problem(N, M, D, seed=N + M, Mg=Mg, ell=ell, u_scale=us)
    if D == 1:
        X = X * 10.0
        p['Zf'] *= 10.0
        p['Zg'] *= 10.0
    engine.set_chunk(chunk or 16384)
    engine.set_data(X, Y)
    scale = 1.7
    ed, kl, g = engine.elbo(p, jitter=1e-6, scale=scale, g_offset=0.0)
    elbo_r, data_r, kl_r, g_r = ot.elbo_and_grad(X, Y, p, 1e-6, scale=scale, chunk=1000)
    c = _cond(p, 1e-6)
    print('cond(Kuu)=%.2e elbo %.10e ref %.10e' % (c, ed - kl, elbo_r))
    assert abs(ed - scale * data_r) <= 1e-7 * abs(scale * data_r)
    assert abs(kl - kl_r) <= 1e-8 * abs(kl_r)
    assert abs((ed - kl) - elbo_r) <= 1e-7 * abs(elbo_r)
    for k in ot.PARAM_KEYS:
        a, b = np.asarray(g[k]).reshape(-1), np.asarray(g_r[k]).reshape(-1)
        e = np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)
        print('  grad %-10s relerr %.2e (max |ref| %.3e)' % (k, e, np.max(np.abs(b))))
        assert e < max(1e-6, 1e-13 * c), (k, e)


def test_inputs_far_from_the_origin_translation_invariance(engine):
    """The RBF kernel only sees differences (KernSE.K, onofftf/main.py:41-57): moving X and Z by the same vector must not change the step
    beyond the rounding of the scaled coordinates (x / ell is formed before the difference, as the reference does: (shift / ell) eps ~ 4e-12
    per difference here, ~1e-8 after the conditioning of the step -- the same for every gradient block).  Code that works with absolute
    coordinates would lose (shift / ell)^2 eps ~ 3e-8 per term instead: k_kgrad's moment sums are taken about the mean inducing input and
    moved to z_m afterwards, and the Z / lengthscale gradients must come out no worse than the blocks that never see a coordinate."""
    X, Y, p = make_problem(3000, 200, 3, seed=21, Mg=136, ell=0.25)
    q = lambda a: np.round(a * 2.0 ** 20) / 2.0 ** 20
    X = q(X); p['Zf'] = q(p['Zf']); p['Zg'] = q(p['Zg'])
    engine.set_chunk(1024)
    engine.set_data(X, Y)
    ed0, kl0, g0 = engine.elbo(p, jitter=1e-6)
    shift = np.array([4096.0, -4096.0, 1024.0])
    p2 = dict(p, Zf=p['Zf'] + shift, Zg=p['Zg'] + shift)
    assert np.array_equal((p2['Zf'] - shift), p['Zf'])          # exactly representable
    engine.set_data(X + shift, Y)
    ed1, kl1, g1 = engine.elbo(p2, jitter=1e-6)
    print('  elbo_data relerr %.2e, kl relerr %.2e' % (abs(ed1 - ed0) / abs(ed0), abs(kl1 - kl0) / abs(kl0)))
    assert abs(ed1 - ed0) < 1e-7 * abs(ed0) and abs(kl1 - kl0) < 1e-7 * abs(kl0)
    for k in g0:
        e = relerr(g1[k], g0[k])
        print('  grad %-10s relerr %.2e' % (k, e))
        assert e < 1e-6, (k, e)


def test_value_only_and_no_kl(engine):
    import zigp_oracle as o
    X, Y, p = make_problem(1000, 64, 3, seed=5)
    engine.set_chunk(16384)
    engine.set_data(X, Y)
    ed, kl, g = engine.elbo(p, need_grad=False)
    assert g is None
    e_r, d_r, klf, klg = o.elbo(X, Y, p, 1e-6)
    assert abs(ed - d_r) < 1e-8 * abs(d_r) and abs(kl - (klf + klg)) < 1e-9 * abs(klf + klg)
    ed2, kl2, _ = engine.elbo(p, include_kl=False, need_grad=False)
    assert kl2 == 0.0 and ed2 == ed
    assert np.allclose(engine.prior_kl(p), [klf, klg], rtol=1e-9)
    # value-only and predict passes over several chunks: with the stream overlap the next chunk's Kuf panels are built on the side stream
    # right behind this chunk's A1 (r6) -- the same kernels in another order of independent work, so the results are the same bits
    X3, Y3, p3 = make_problem(5000, 150, 3, seed=6, Mg=70)
    engine.set_chunk(1024)                                  # five passes, the last one partial
    engine.set_data(X3, Y3)
    on = (engine.elbo(p3, need_grad=False), engine.predict(p3, X3))
    engine.set_overlap(False)
    off = (engine.elbo(p3, need_grad=False), engine.predict(p3, X3))
    engine.set_overlap(True)
    assert on[0][0] == off[0][0] and on[0][1] == off[0][1] and np.array_equal(on[1], off[1])
    e3 = o.elbo(X3, Y3, p3, 1e-6)
    assert abs(on[0][0] - e3[1]) < 1e-8 * abs(e3[1])
    engine.set_chunk(16384)


def test_row_shards_sum_to_full(engine):
    """Data-parallel invariance: shard partials + KL once == full-batch step."""
    X, Y, p = make_problem(4096, 128, 3, seed=11)
    engine.set_chunk(1024)
    engine.set_data(X, Y)
    ed, kl, g = engine.elbo(p)
    parts = [engine.elbo(p, rows=(0, 1500), include_kl=True), engine.elbo(p, rows=(1500, 4096), include_kl=False)]
    assert abs(sum(q[0] for q in parts) - ed) < 1e-11 * abs(ed)
    assert parts[0][1] == kl and parts[1][1] == 0.0
    for k in g:
        s = np.asarray(parts[0][2][k]) + np.asarray(parts[1][2][k])
        assert np.max(np.abs(s - np.asarray(g[k]))) <= 1e-8 * max(np.max(np.abs(np.asarray(g[k]))), 1e-300), k   # summation order x cond(Kuu)


def test_bit_stable_run_to_run(engine):
    X, Y, p = make_problem(3000, 128, 3, seed=3)
    engine.set_chunk(1024)
    engine.set_data(X, Y)
    a = engine.elbo(p)
    b = engine.elbo(p)
    assert a[0] == b[0] and a[1] == b[1]
    for k in a[2]:
        assert np.array_equal(np.asarray(a[2][k]), np.asarray(b[2][k]))


def test_bit_stable_over_many_calls_with_other_row_ranges_in_between():
    """The three-stream step (factorisation chains on two streams, buffers / zeroed accumulators / first Kuf panels on a third, side kernels
    beside the rank-N updates): 40 calls on a fresh engine under the library's own chunk rule, every third one on a different row range (the
    accumulators are re-zeroed, the panels re-sized) -- every full-range result equals the first bit for bit."""
    import zigp
    e = zigp.DenseEngine(0)
    try:
        X, Y, p = make_problem(30000, 200, 3, seed=5)
        e.set_data(X, Y)
        ref = e.elbo(p)
        for i in range(40):
            if i % 3 == 1:
                e.elbo(p, rows=(1000, 17000), need_grad=bool(i % 2))
                continue
            out = e.elbo(p)
            assert out[0] == ref[0] and out[1] == ref[1]
            for k in ref[2]:
                assert np.array_equal(np.asarray(out[2][k]), np.asarray(ref[2][k])), (i, k)
    finally:
        e.close()


def test_not_pd_raises(engine):
    import zigp
    X, Y, p = make_problem(500, 32, 3, seed=1)
    p['Zf'][5, 0] = np.nan   # NaN pivot: tf.cholesky fails on such a Kuu too
    engine.set_data(X, Y)
    with pytest.raises(zigp.NotPositiveDefiniteError):
        engine.elbo(p, jitter=0.0)
    p['Zf'][5, 0] = 0.5       # the context recovers
    assert np.isfinite(engine.elbo(p, need_grad=False)[0])


def test_bad_arguments(engine):
    X, Y, p = make_problem(100, 16, 3, seed=1)
    engine.set_data(X, Y)
    q = dict(p)
    q['var_f'] = -1.0
    with pytest.raises(ValueError):
        engine.elbo(q)
    with pytest.raises(ValueError):
        engine.elbo(p, rows=(0, 101))
    q = dict(p)
    q['u_fs_sqrt'] = p['u_fs_sqrt'].copy()
    q['u_fs_sqrt'][3] = 0.0
    with pytest.raises(ValueError):
        engine.elbo(q)
    with pytest.raises(ValueError):
        engine.set_chunk(1000)


def _longdouble_conditional(Xnew, Z, ell, var, q_mu, q_sqrt, jitter):
    """80-bit evaluation of the same formulas (onofftf/main.py:257-305) -- 'truth' for the ill-conditioned case."""
    ld = np.longdouble
    Xn, Zs = (Xnew / ell).astype(ld), (Z / ell).astype(ld)
    def K(A, B):
        d2 = ((A[:, None, :] - B[None, :, :]) ** 2).sum(-1)
        return ld(var) * np.exp(-d2 / 2)
    M = Z.shape[0]
    Kmm = K(Zs, Zs) + ld(jitter) * np.eye(M, dtype=ld)
    Kmn = K(Zs, Xn)
    L = np.zeros((M, M), dtype=ld)
    for j in range(M):
        L[j, j] = np.sqrt(Kmm[j, j] - (L[j, :j] ** 2).sum())
        L[j + 1:, j] = (Kmm[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    A = np.zeros_like(Kmn)
    for i in range(M):
        A[i] = (Kmn[i] - L[i, :i] @ A[:i]) / L[i, i]
    fvar = ld(var) - (A ** 2).sum(0)
    B = np.zeros_like(A)
    for i in range(M - 1, -1, -1):
        B[i] = (A[i] - L[i + 1:, i] @ B[i + 1:]) / L[i, i]
    fmean = B.T @ q_mu.reshape(-1).astype(ld)
    fvar = fvar + ((B * q_sqrt.reshape(-1, 1).astype(ld)) ** 2).sum(0)
    return fmean, fvar


def test_ill_conditioned_accuracy_vs_extended_precision(engine):
    """cond(Kuu) ~ 2e7 with a rough u_m: GPU and oracle both drift from 80-bit truth; the GPU W-form
    (explicit triangular inverse + GEMMs) must not be meaningfully worse than the oracle's triangular solves."""
    import zigp_oracle as o
    X, Y, p = make_problem(450, 50, 1, seed=500, ell=2.0, u_scale=0.5)
    X, p['Zf'], p['Zg'] = X * 10.0, p['Zf'] * 10.0, p['Zg'] * 10.0
    tm, tv = _longdouble_conditional(X, p['Zf'], p['ell_f'], p['var_f'], p['u_fm'], p['u_fs_sqrt'], 1e-6)
    out = engine.predict(p, X, jitter=1e-6)
    ref = o.build_predict(X, p, 1e-6)
    e_gpu_m, e_orc_m = relerr(out[3], tm.astype(np.float64)), relerr(ref[3].reshape(-1), tm.astype(np.float64))
    e_gpu_v, e_orc_v = relerr(out[4], tv.astype(np.float64)), relerr(ref[4].reshape(-1), tv.astype(np.float64))
    print('fmean: gpu %.2e oracle %.2e ; fvar: gpu %.2e oracle %.2e' % (e_gpu_m, e_orc_m, e_gpu_v, e_orc_v))
    assert e_gpu_m < 10 * e_orc_m + 1e-9 and e_gpu_v < 10 * e_orc_v + 1e-9


def test_wide_input_dimension_and_large_M(engine):
    """D = 5 (ARD) and M = 1100 (nine 128-blocks, ragged) against the oracle."""
    import zigp_oracle_torch as ot
    for (N, M, D, ell) in ((1200, 70, 5, 0.8), (1300, 1100, 3, 0.12)):
        X, Y, p = make_problem(N, M, D, seed=N + D, ell=ell, u_scale=0.1)
        engine.set_chunk(32768)
        engine.set_data(X, Y)
        ed, kl, g = engine.elbo(p, jitter=1e-6)
        e_r, d_r, kl_r, g_r = ot.elbo_and_grad(X, Y, p, 1e-6, chunk=700)
        c = _cond(p, 1e-6)
        print('N=%d M=%d D=%d cond %.1e elbo rel %.1e' % (N, M, D, c, abs((ed - kl) - e_r) / abs(e_r)))
        assert abs((ed - kl) - e_r) <= 1e-7 * abs(e_r)
        for k in ot.PARAM_KEYS:
            a, b = np.asarray(g[k]).reshape(-1), np.asarray(g_r[k]).reshape(-1)
            assert np.max(np.abs(a - b)) <= max(1e-6, 1e-13 * c) * max(np.max(np.abs(b)), 1e-300), k


def test_empty_row_range_gives_minus_kl_gradient(engine):
    """rows = (k, k): no data term; ELBO = -KL and the gradient is -dKL (what a rank without rows contributes)."""
    import zigp_oracle_torch as ot
    X, Y, p = make_problem(600, 40, 3, seed=8)
    engine.set_data(X, Y)
    ed, kl, g = engine.elbo(p, rows=(100, 100))
    assert ed == 0.0
    P = ot.make_leaves(p)
    k = ot.prior_kl(P, 1e-6)
    (-k).backward()
    assert abs(kl - float(k.detach())) <= 1e-9 * abs(kl)
    for key in ('Zf', 'u_fm', 'u_gs_sqrt', 'ell_g', 'var_f'):
        a, b = np.asarray(g[key]).reshape(-1), P[key].grad.numpy().reshape(-1)
        assert np.max(np.abs(a - b)) <= 1e-7 * max(np.max(np.abs(b)), 1e-300), key
    assert g['noise'] == 0.0


# ---- mean function of f (onoffgpf/OnOffSVGP.py:29,134): Zero / Constant / Linear as m(x) = b + a . x ------------------
@pytest.mark.parametrize('kind', ['constant', 'linear'])
def test_mean_function_matches_oracle(engine, kind):
    import zigp_oracle as o
    import zigp_oracle_torch as ot
    N, M, D = 3000, 96, 3
    X, Y, p = make_problem(N, M, D, seed=21, ell=0.35)
    p = dict(p, mean_b=0.37)
    if kind == 'linear':
        p['mean_a'] = np.array([0.5, -0.25, 0.125])
    engine.set_chunk(1024)            # several chunks: the per-block partial sums accumulate across them
    engine.set_data(X, Y)
    ed, kl, g = engine.elbo(p, jitter=1e-6, scale=1.7)
    e_r, d_r, kl_r, g_r = ot.elbo_and_grad(X, Y, p, 1e-6, scale=1.7)
    assert abs(ed - 1.7 * d_r) <= 1e-8 * abs(1.7 * d_r) and abs(kl - kl_r) <= 1e-8 * abs(kl_r)
    keys = ['Zf', 'u_fm', 'u_gm', 'u_fs_sqrt', 'ell_f', 'var_f', 'noise', 'mean_b'] + (['mean_a'] if kind == 'linear' else [])
    for k in keys:
        a, b = np.asarray(g[k], dtype=float).reshape(-1), np.asarray(g_r[k], dtype=float).reshape(-1)
        e = np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)
        print('mean function %s: grad %s relerr %.2e' % (kind, k, e))
        assert e < 1e-6, (k, e)       # tolerance 1e-6 relative, fp64
    out = engine.predict(p, X[:500], jitter=1e-6)
    ref = o.build_predict(X[:500], p, 1e-6)
    for i in range(9):
        assert relerr(out[i], ref[i].reshape(-1)) < 1e-7, i
    # the offset really is in: without the mean function the value differs, and the setting does not leak into the next call
    p0 = {k: v for k, v in p.items() if not k.startswith('mean_')}
    ed0, _, g0 = engine.elbo(p0, jitter=1e-6, scale=1.7)
    assert abs(ed0 - ed) > 1e-3 * abs(ed) and 'mean_b' not in g0
    _, d0_r, _, _ = ot.elbo_and_grad(X, Y, p0, 1e-6, scale=1.7, need_grad=False)
    assert abs(ed0 - 1.7 * d0_r) <= 1e-8 * abs(1.7 * d0_r)
    engine.set_chunk(32768)


def test_constant_mean_at_zero_still_gets_its_gradient(engine):
    """GPflow's Constant() defaults to c = 0 and an optimiser may land on 0: 'enabled' must not depend on the value."""
    import zigp_oracle_torch as ot
    X, Y, p = make_problem(1500, 48, 2, seed=33, ell=0.4)
    engine.set_data(X, Y)
    p = dict(p, mean_b=0.0)
    ed, kl, g = engine.elbo(p, jitter=1e-6)
    e_r, d_r, kl_r, g_r = ot.elbo_and_grad(X, Y, p, 1e-6)
    assert abs(ed - d_r) <= 1e-8 * abs(d_r)
    assert abs(g_r['mean_b']) > 1e-3 and abs(g['mean_b'] - g_r['mean_b']) <= 1e-6 * abs(g_r['mean_b'])   # tolerance 1e-6, fp64
    p = dict(p, mean_a=np.zeros(2))
    ed, kl, g = engine.elbo(p, jitter=1e-6)
    e_r, d_r, kl_r, g_r = ot.elbo_and_grad(X, Y, p, 1e-6)
    assert np.max(np.abs(np.asarray(g['mean_a']) - np.asarray(g_r['mean_a']))) <= 1e-6 * np.max(np.abs(np.asarray(g_r['mean_a'])))


def test_mean_function_bad_arguments(engine):
    X, Y, p = make_problem(200, 16, 2, seed=3)
    engine.set_data(X, Y)
    with pytest.raises(ValueError):
        engine.elbo(dict(p, mean_a=np.ones(3)))
    with pytest.raises(ValueError):
        engine.elbo(dict(p, mean_b=float('nan')))


def test_stream_overlap_is_bit_identical(engine):
    """the side-stream schedule (kgrad / next Kuf panel under the SYRKs) only reorders independent kernels"""
    X, Y, p = make_problem(9000, 130, 3, seed=8)
    engine.set_chunk(2048)            # 5 chunks, the last one partial
    engine.set_data(X, Y)
    out = {}
    for on in (True, False, True):
        engine.set_overlap(on)
        for prof in (False, True):    # timed chunks fall back to the single stream inside an overlapped step
            engine.profile_enable(prof)
            out[(on, prof)] = engine.elbo(p, jitter=1e-6, scale=1.3)
    engine.profile_enable(False)
    engine.set_overlap(True)          # the library default
    engine.set_chunk(32768)
    ref = out[(False, False)]
    for k, (ed, kl, g) in out.items():
        assert ed == ref[0] and kl == ref[1], k
        for name in g:
            assert np.array_equal(np.asarray(g[name]), np.asarray(ref[2][name])), (k, name)


def test_mean_function_shards_sum_to_full(engine):
    """per-shard partial gradients of the mean function add up (what ShardedELBO all-reduces)"""
    X, Y, p = make_problem(5000, 64, 2, seed=4)
    p = dict(p, mean_a=np.array([0.3, -0.2]), mean_b=0.1)
    engine.set_chunk(2048)
    engine.set_data(X, Y)
    ed, kl, g = engine.elbo(p)
    parts = [engine.elbo(p, rows=r, include_kl=(i == 0)) for i, r in enumerate(((0, 1777), (1777, 3100), (3100, 5000)))]
    engine.set_chunk(32768)
    assert abs(sum(q[0] for q in parts) - ed) <= 1e-11 * abs(ed)
    for k in ('mean_a', 'mean_b', 'u_fm', 'noise'):
        s = sum(np.asarray(q[2][k], dtype=float) for q in parts)
        assert np.max(np.abs(s - np.asarray(g[k], dtype=float))) <= 1e-9 * max(np.max(np.abs(np.asarray(g[k], dtype=float))), 1e-300), k


def test_select_rows_equals_uploading_the_gathered_minibatch(engine):
    """zigp_select_rows gathers a row-index sample (repeats allowed) of the RESIDENT data on the device -- MinibatchData's per-step sample
    (onoffgpf/OnOffSVGP.py:46-47) without re-uploading X, Y: same numbers, bit for bit, as uploading X[idx], Y[idx]."""
    X, Y, p = make_problem(5000, 70, 3, seed=12)
    engine.set_chunk(2048)
    engine.set_data(X, Y)
    full = engine.elbo(p, jitter=1e-6)
    idx = np.random.RandomState(3).randint(5000, size=1700)
    engine.select_rows(idx)
    a = engine.elbo(p, jitter=1e-6, scale=5000 / 1700.0)
    assert engine.N == 1700
    engine.select_rows(None)
    again = engine.elbo(p, jitter=1e-6)
    engine.set_data(X[idx], Y[idx])
    b = engine.elbo(p, jitter=1e-6, scale=5000 / 1700.0)
    assert a[0] == b[0] and a[1] == b[1] and full[0] == again[0]
    for k in a[2]:
        assert np.array_equal(np.asarray(a[2][k]), np.asarray(b[2][k])), k
    engine.set_data(X, Y)
    with pytest.raises(ValueError):
        engine.select_rows([0, 5000])
    engine.set_chunk(32768)


def test_chunk_rule_is_reported_by_the_library():
    """zigp_get_chunk: the rows-per-pass rule lives in the library only (bench.py reads it instead of re-implementing it)"""
    import zigp
    e = zigp.DenseEngine(0)
    try:
        assert [e.get_chunk(M) for M in (1024, 1000, 512, 256, 128, 50, 2048)] == [32768, 32768, 65536, 131072, 131072, 131072, 32768]
        # rows per pass of a given row range: equal passes; a short range in ONE pass while its panels fit 9 GB (M = 2048 and up
        # fall back to the M-scaled chunk)
        assert e.get_chunk_rows(1024, 1000000) == 32768 and e.get_chunk_rows(512, 100000) == 100352 and e.get_chunk_rows(1024, 125000) == 125952
        assert e.get_chunk_rows(1024, 131072) == 131072 and e.get_chunk_rows(2048, 131072) == 32768 and e.get_chunk_rows(2048, 40000) == 40960
        assert e.get_chunk_rows(1024, 0) == 1024
        e.set_chunk(4096)
        assert e.get_chunk(1024) == 4096 and e.get_chunk(64) == 4096 and e.get_chunk_rows(512, 100000) == 4096
        with pytest.raises(ValueError):
            e.get_chunk(0)
        assert e.comm_info() == dict(rank=0, nranks=0, allreduce_calls=0)
        with pytest.raises(ValueError):
            e.comm_init(2, 2, b'\0' * 128)          # rank out of range: rejected before anything touches RCCL
        with pytest.raises(ValueError):
            e.comm_init(0, 1, b'short')
    finally:
        e.close()


def test_automatic_rule_takes_a_short_row_range_in_one_pass_with_the_same_result():
    """Under the library's own chunk rule (a fresh engine: no zigp_set_chunk) a row range of <= 131072 rows is ONE pass; a long range is cut
    at the M-scaled chunk.  Both must give what explicit small chunks give (value 1e-12; gradients 1e-6 relative, the tolerance of every gradient comparison here -- measured
    9e-10 on Z and 3e-8 on the kernel variance, a sum with cancellation: the partial sums are split differently and the reverse M x M stage amplifies that by cond(Kuu)), and the rule must switch at the boundary without a seam: 131072 rows in one pass against 131073 in two."""
    import zigp
    e = zigp.DenseEngine(0)
    try:
        N, M, D = 40000, 130, 2
        X, Y, p = make_problem(N, M, D, seed=31, ell=0.4)
        e.set_data(X, Y)
        ed_a, kl_a, g_a = e.elbo(p, jitter=1e-6)                      # automatic: one pass of 40960 columns
        ed_r, _, _ = e.elbo(p, jitter=1e-6, rows=(100, 35000))        # a row range through the same rule
        e.set_chunk(8192)
        ed_c, kl_c, g_c = e.elbo(p, jitter=1e-6)                      # five chunks
        ed_rc, _, _ = e.elbo(p, jitter=1e-6, rows=(100, 35000))
        assert abs(ed_a - ed_c) <= 1e-12 * abs(ed_c) and kl_a == kl_c and abs(ed_r - ed_rc) <= 1e-12 * abs(ed_rc)
        for k in ('Zf', 'Zg', 'u_fm', 'u_gs_sqrt', 'ell_f', 'var_g', 'noise'):
            a, b = np.asarray(g_a[k], dtype=float).reshape(-1), np.asarray(g_c[k], dtype=float).reshape(-1)
            assert np.max(np.abs(a - b)) <= 1e-6 * np.max(np.abs(b)), k
    finally:
        e.close()
    e = zigp.DenseEngine(0)
    try:
        N, M, D = 131073, 64, 2
        X, Y, p = make_problem(N, M, D, seed=32, ell=0.5)
        e.set_data(X, Y)
        one, _, _ = e.elbo(p, jitter=1e-6, rows=(0, 131072), need_grad=False)        # one pass
        last, _, _ = e.elbo(p, jitter=1e-6, rows=(131072, 131073), include_kl=False, need_grad=False)
        both, _, _ = e.elbo(p, jitter=1e-6, need_grad=False)                         # 131073 rows: the M-scaled chunk (131072 at M = 64) -> two passes
        assert abs((one + last) - both) <= 1e-12 * abs(both)
    finally:
        e.close()


def test_M2048_sixteen_row_blocks_and_the_chunk_rule_beyond_its_one_pass_bound(engine):
    """M = 2048 (16 row blocks of 128; Mg = 1100: 9 blocks): the tile lists, the split-K plan of the rank-N update and the blocked
    factorisation at twice cfg3's depth, against the oracle; then a row range beyond the one-pass bound of the chunk rule (4 panels of
    8 Mp span bytes per latent <= 9 GB, include/zigp.h zigp_get_chunk_rows): 80 000 rows at M = 2048 go through in three passes of the
    M-scaled chunk -- same step as with a fixed small chunk, and a slice of it against the oracle."""
    import zigp_oracle as o
    import zigp_oracle_torch as ot
    engine.set_chunk(0)                                         # the automatic rule (earlier tests of the session fixed the chunk)
    assert engine.get_chunk(2048) == 32768
    assert engine.get_chunk_rows(2048, 60000) == 60416          # 4 * 2 * 8 * 2048 * 60416 B = 7.9 GB: one pass
    assert engine.get_chunk_rows(2048, 80000) == 27648          # 10.6 GB: the rule falls back to ceil(80000 / 32768) = 3 equal passes
    X, Y, p = make_problem(80000, 2048, 3, seed=77, Mg=1100, ell=0.1)
    c = _cond(p, 1e-6)
    n = 2500
    engine.set_data(X[:n], Y[:n])
    ed, kl, g = engine.elbo(p, jitter=1e-6)
    e_r, d_r, kl_r, g_r = ot.elbo_and_grad(X[:n], Y[:n], p, 1e-6, chunk=1250)
    print('M=2048: cond(Kuu)=%.2e elbo rel %.2e kl rel %.2e' % (c, abs((ed - kl) - e_r) / abs(e_r), abs(kl - kl_r) / abs(kl_r)))
    assert abs(ed - d_r) <= 1e-7 * abs(d_r) and abs(kl - kl_r) <= 1e-8 * abs(kl_r)
    for k in ot.PARAM_KEYS:
        e = relerr(np.asarray(g[k]).reshape(-1), np.asarray(g_r[k]).reshape(-1))
        print('  M=2048 grad %-10s relerr %.2e' % (k, e))
        assert e < max(1e-6, 1e-13 * c), (k, e)
    out = engine.predict(p, X[:n], jitter=1e-6)
    ref = o.build_predict(X[:n], p, 1e-6)
    for i in range(9):
        assert relerr(out[i], ref[i].reshape(-1)) < max(1e-9, min(1e-6, 1e-13 * c)), i
    # the long range: automatic rule (3 passes of 27648 rows) against 10 passes of 8192
    engine.set_chunk(0)
    engine.set_data(X, Y)
    ed_a, kl_a, g_a = engine.elbo(p, jitter=1e-6)
    engine.set_chunk(8192)
    ed_b, kl_b, g_b = engine.elbo(p, jitter=1e-6)
    assert abs(ed_a - ed_b) <= 1e-10 * abs(ed_a) and kl_a == kl_b
    for k in g_a:
        assert relerr(g_a[k], g_b[k]) <= 1e-7, k
    rows = (41000, 43000)                                        # inside the second pass of the automatic rule
    eds, _, gs = engine.elbo(p, jitter=1e-6, rows=rows, include_kl=False)
    e_r, d_r, kl_r, g_r = ot.elbo_and_grad(X[rows[0]:rows[1]], Y[rows[0]:rows[1]], p, 1e-6, include_kl=False, chunk=1000)
    assert abs(eds - d_r) <= 1e-7 * abs(d_r)
    for k in ot.PARAM_KEYS:
        assert relerr(np.asarray(gs[k]).reshape(-1), np.asarray(g_r[k]).reshape(-1)) < max(1e-6, 1e-13 * c), k
    engine.set_chunk(16384)


@pytest.mark.parametrize('span', [600.0, 5000.0])
def test_inducing_inputs_spanning_many_lengthscales(engine, span):
    """A long 1-D input (a time series): M = 400 inducing points on a grid over `span` lengthscales.  k_kgrad takes its moment sums about
    the MEAN inducing input and shifts them to z_m afterwards, which costs (|z_m - mean| / ell)^2 ulp in the Z / lengthscale gradients
    (ADVICE r5): at a spread of 300 lengthscales that is ~2e-11 and stays the fast path; beyond 1e3 lengthscales from the mean (span 5000:
    2500) the engine switches that latent to per-row differences (k_kgrad<D, true>), so the gradients keep the 1e-6 parity with the oracle
    (autograd of KernSE.K's broadcast difference, onofftf/main.py:41-57) however long the input is."""
    import zigp_oracle_torch as ot
    rs = np.random.RandomState(17)
    N, M = 3000, 400
    X = np.sort(rs.rand(N, 1) * span, axis=0)
    f = np.sin(X[:, 0] / 3.0)
    Y = np.where(np.sin(X[:, 0] / 7.0) + 0.5 * rs.randn(N) > 0, f + 0.1 * rs.randn(N), 0.0)[:, None]
    Z = (np.arange(M)[:, None] + 0.5) * (span / M)
    p = dict(Zf=Z, Zg=Z + 0.25 * span / M, u_fm=0.3 * rs.randn(M, 1), u_gm=0.3 * rs.randn(M, 1), u_fs_sqrt=0.3 + rs.rand(M, 1),
             u_gs_sqrt=0.3 + rs.rand(M, 1), ell_f=np.array([1.0]), ell_g=np.array([1.2]), var_f=1.0, var_g=5.0, noise=0.01)
    engine.set_chunk(1024)
    engine.set_data(X, Y)
    ed, kl, g = engine.elbo(p, jitter=1e-6)
    e_r, d_r, kl_r, g_r = ot.elbo_and_grad(X, Y, p, 1e-6, chunk=1000)
    c = _cond(p, 1e-6)
    assert abs(ed - d_r) <= 1e-7 * abs(d_r) and abs(kl - kl_r) <= 1e-8 * abs(kl_r)
    for k in ot.PARAM_KEYS:
        e = relerr(np.asarray(g[k]).reshape(-1), np.asarray(g_r[k]).reshape(-1))
        print('  span %g ell: cond %.1e grad %-10s relerr %.2e' % (span, c, k, e))
        assert e < max(1e-6, 1e-13 * c), (k, e)
    engine.set_chunk(16384)
