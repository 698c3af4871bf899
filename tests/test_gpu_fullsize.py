"""Full-size (BASELINE.json cfg3: N = 1e6, M = 1024, D = 3) properties that need no oracle at that size, plus an oracle
check on a slice.  Same synthetic generator as bench.py."""
import os
import sys

import numpy as np
import pytest

from conftest import make_problem, relerr

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def cfg3(engine):
    sys.path.insert(0, ROOT)
    import bench
    import torch
    X, Y, p = bench.synth(1000000, 1024, 3)
    Xd, Yd = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()
    engine.set_chunk(32768)
    engine.set_data_device(Xd, Yd)          # device-resident inputs, as in bench.py
    return X, Y, p


def test_cfg3_chunk_size_invariance_and_shard_additivity(engine, cfg3):
    X, Y, p = cfg3
    N = X.shape[0]
    ed, kl, g = engine.elbo(p)
    engine.set_chunk(16384)
    ed2, kl2, g2 = engine.elbo(p)
    engine.set_chunk(32768)
    assert abs(ed - ed2) <= 1e-11 * abs(ed) and kl == kl2
    for k in g:
        assert np.max(np.abs(np.asarray(g[k]) - np.asarray(g2[k]))) <= 1e-8 * max(np.max(np.abs(np.asarray(g[k]))), 1e-300), k
    # three ragged shards + KL once == full batch (the 8-GPU layout in miniature)
    cuts = [0, 333333, 700001, N]
    parts = [engine.elbo(p, rows=(cuts[i], cuts[i + 1]), include_kl=(i == 0)) for i in range(3)]
    assert abs(sum(q[0] for q in parts) - ed) <= 1e-11 * abs(ed)
    assert parts[0][1] == kl and parts[1][1] == 0.0
    for k in g:
        s = sum(np.asarray(q[2][k]) for q in parts)
        assert np.max(np.abs(s - np.asarray(g[k]))) <= 1e-8 * max(np.max(np.abs(np.asarray(g[k]))), 1e-300), k


def test_cfg3_scale_linearity_and_value_only(engine, cfg3):
    X, Y, p = cfg3
    ed1, kl1, g1 = engine.elbo(p, scale=1.0, rows=(0, 200000))
    ed3, kl3, g3 = engine.elbo(p, scale=3.0, rows=(0, 200000))
    assert abs(ed3 - 3.0 * ed1) <= 1e-12 * abs(ed3) and kl1 == kl3
    edv, klv, gv = engine.elbo(p, rows=(0, 200000), need_grad=False)
    assert gv is None and abs(edv - ed1) <= 1e-13 * abs(ed1) and klv == kl1
    # data-term gradient scales, KL gradient does not:  g3 - g1 = 2 * (g1 + dKL)  ->  check on the noise variance (no KL part)
    assert abs(g3['noise'] - 3.0 * g1['noise']) <= 1e-10 * abs(g3['noise'])


def test_cfg3_slice_matches_oracle(engine, cfg3):
    import zigp_oracle_torch as ot
    X, Y, p = cfg3
    n = 6000
    ed, kl, g = engine.elbo(p, rows=(500000, 500000 + n))
    e_r, d_r, kl_r, g_r = ot.elbo_and_grad(X[500000:500000 + n], Y[500000:500000 + n], p, 1e-6, chunk=3000)
    print('cfg3 slice: data rel %.2e kl rel %.2e' % (abs(ed - d_r) / abs(d_r), abs(kl - kl_r) / abs(kl_r)))
    assert abs(ed - d_r) <= 1e-8 * abs(d_r) and abs(kl - kl_r) <= 1e-9 * abs(kl_r)
    for k in ot.PARAM_KEYS:
        a, b = np.asarray(g[k]).reshape(-1), np.asarray(g_r[k]).reshape(-1)
        assert np.max(np.abs(a - b)) <= 1e-6 * max(np.max(np.abs(b)), 1e-300), k


def test_cfg3_predict_first_rows_match_oracle(engine, cfg3):
    import zigp_oracle as o
    X, Y, p = cfg3
    out = engine.predict(p, X[:5000])
    ref = o.build_predict(X[:5000], p, 1e-6)
    for i in range(9):
        r = ref[i].reshape(-1)
        assert np.max(np.abs(out[i] - r)) <= 1e-7 * max(np.max(np.abs(r)), 1e-300), i


def test_cfg3_stress_lengthscale_accuracy(engine):
    """SURVEY.md §8d stress variant: cfg3 geometry with lengthscale 0.2 (cond(Kuu) ~ 1e8 at M = 1024, jitter 1e-6).  GPU and
    oracle are both judged against an 80-bit evaluation of GPConditional on a few points (the GPU must stay within 10x of the
    oracle's own error), and the ELBO on a slice must still agree to 1e-6."""
    sys.path.insert(0, ROOT)
    import bench
    import zigp_oracle as o
    import zigp_oracle_torch as ot
    from test_gpu_dense import _longdouble_conditional, _cond
    X, Y, p = bench.synth(4096, 1024, 3)
    p = dict(p, ell_f=np.full(3, 0.2), ell_g=np.full(3, 0.2))
    c = _cond(p, 1e-6)
    npts = 24
    out = engine.predict(p, X[:npts], jitter=1e-6)
    ref = o.build_predict(X[:npts], p, 1e-6)
    for tag, im, iv in (('f', 3, 4), ('g', 5, 6)):
        tm, tv = _longdouble_conditional(X[:npts], p['Z' + tag], p['ell_' + tag], p['var_' + tag], p['u_%sm' % tag], p['u_%ss_sqrt' % tag], 1e-6)
        tm, tv = tm.astype(np.float64), tv.astype(np.float64)
        for nm, idx, truth in (('mean', im, tm), ('var', iv, tv)):
            e_gpu = np.max(np.abs(out[idx] - truth)) / np.max(np.abs(truth))
            e_orc = np.max(np.abs(ref[idx].reshape(-1) - truth)) / np.max(np.abs(truth))
            print('stress cond %.1e %s%s: gpu vs 80-bit %.2e, oracle vs 80-bit %.2e' % (c, tag, nm, e_gpu, e_orc))
            assert e_gpu < max(1e-6, 10 * e_orc), (tag, nm, e_gpu, e_orc)
    engine.set_chunk(32768)
    engine.set_data(X, Y)
    ed, kl, g = engine.elbo(p, jitter=1e-6)
    e_r, d_r, kl_r, g_r = ot.elbo_and_grad(X, Y, p, 1e-6, chunk=2048)
    print('stress elbo rel %.2e kl rel %.2e' % (abs((ed - kl) - e_r) / abs(e_r), abs(kl - kl_r) / abs(kl_r)))
    assert abs((ed - kl) - e_r) <= 1e-6 * abs(e_r)      # north-star tolerance
    for k in ('u_fm', 'u_fs_sqrt', 'noise', 'var_f'):
        a, b = np.asarray(g[k]).reshape(-1), np.asarray(g_r[k]).reshape(-1)
        e = np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)
        print('  stress grad %s relerr %.2e' % (k, e))
        assert e <= max(1e-6, 1e-13 * c), (k, e)


# ---- BASELINE.json configs[3]: N = 8e6 over 8 ranks of 1e6 rows -- every shard of it through ONE GPU -------------------------
def test_cfg4_all_eight_shards_sum_to_one_8e6_row_call(engine):
    """cfg4 exactly as bench.py --gpus 8 partitions it (bench.weak_shard(1e6, 1024, 3, r, 8), r = 0..7; KL on rank 0 only), at full size:
    the eight per-rank results, summed as the all-reduce sums them, against ONE resident 8e6-row call on the same GPU.  The data term is
    a plain sum over points (onoffgpf/OnOffSVGP.py:119-122, scripts/onoff.py:307), so the split is exact up to summation order: value
    1e-11, gradients 1e-8 relative.  What this leaves open about cfg4 is RCCL itself, nothing else."""
    sys.path.insert(0, ROOT)
    import bench
    import torch
    R, n = bench.CFG4_SHARDS, 1000000
    X8, Y8, p = bench.synth(R * n, 1024, 3)                      # the cfg4 stream in one piece (SURVEY 8d generator over 8e6 rows)
    for r in (0, 5, 7):                                          # rank r's shard as the launcher draws it == rows [r n, (r + 1) n) of the stream
        Xr, Yr, pr = bench.weak_shard(n, 1024, 3, r, R)
        assert np.array_equal(Xr, X8[r * n:(r + 1) * n]) and np.array_equal(Yr, Y8[r * n:(r + 1) * n])
        assert all(np.array_equal(np.asarray(pr[k]), np.asarray(p[k])) for k in p)
        del Xr, Yr
    Xd, Yd = torch.from_numpy(X8).cuda(), torch.from_numpy(Y8).cuda()
    engine.set_chunk(32768)
    parts = []
    for r in range(R):                                           # one "rank" after the other: its own resident shard, its own call
        engine.set_data_device(Xd[r * n:(r + 1) * n], Yd[r * n:(r + 1) * n])
        parts.append(engine.elbo(p, include_kl=(r == 0)))
    engine.set_data_device(Xd, Yd)
    ed, kl, g = engine.elbo(p)                                   # the whole of cfg4 in one call
    sd = sum(q[0] for q in parts)
    print('cfg4: elbo_data %.12e, sum of 8 shards rel %.2e' % (ed, abs(sd - ed) / abs(ed)))
    assert abs(sd - ed) <= 1e-11 * abs(ed)
    assert parts[0][1] == kl and all(q[1] == 0.0 for q in parts[1:])
    for k in g:
        s = sum(np.asarray(q[2][k]) for q in parts)
        e = np.max(np.abs(s - np.asarray(g[k]))) / max(np.max(np.abs(np.asarray(g[k]))), 1e-300)
        print('  cfg4 grad %s: sum of 8 shards rel %.2e' % (k, e))
        assert e <= 1e-8, (k, e)
    # the same partition by row ranges of the resident 8e6 rows (what --scaling strong does), two of the eight
    for r in (3, 7):
        er, _, gr = engine.elbo(p, rows=(r * n, (r + 1) * n), include_kl=False)
        assert abs(er - parts[r][0]) <= 1e-12 * abs(er)
        assert all(relerr(gr[k], parts[r][2][k]) <= 1e-10 for k in gr)
    engine.set_data(X8[:1024], Y8[:1024])                        # drop the references to the 8e6-row tensors
    del Xd, Yd
    torch.cuda.empty_cache()


# ---- BASELINE.json configs[1]: N = 1e5, D = 3, M = 512 ---------------------------------------------------------------------
@pytest.fixture(scope='module')
def cfg2(engine):
    sys.path.insert(0, ROOT)
    import bench
    X, Y, p = bench.synth(100000, 512, 3)
    return X, Y, p


def test_cfg2_chunk_invariance_shard_additivity_and_slice_vs_oracle(engine, cfg2):
    import zigp_oracle_torch as ot
    X, Y, p = cfg2
    N = X.shape[0]
    engine.set_chunk(32768)
    engine.set_data(X, Y)
    ed, kl, g = engine.elbo(p)
    engine.set_chunk(8192)
    ed2, kl2, g2 = engine.elbo(p)
    engine.set_chunk(32768)
    assert abs(ed - ed2) <= 1e-11 * abs(ed) and kl == kl2
    for k in g:
        assert np.max(np.abs(np.asarray(g[k]) - np.asarray(g2[k]))) <= 1e-8 * max(np.max(np.abs(np.asarray(g[k]))), 1e-300), k
    cuts = [0, 33333, 70001, N]
    parts = [engine.elbo(p, rows=(cuts[i], cuts[i + 1]), include_kl=(i == 0)) for i in range(3)]
    assert abs(sum(q[0] for q in parts) - ed) <= 1e-11 * abs(ed)
    for k in g:
        s = sum(np.asarray(q[2][k]) for q in parts)
        assert np.max(np.abs(s - np.asarray(g[k]))) <= 1e-8 * max(np.max(np.abs(np.asarray(g[k]))), 1e-300), k
    n = 6000
    eds, kls, gs = engine.elbo(p, rows=(50000, 50000 + n))
    e_r, d_r, kl_r, g_r = ot.elbo_and_grad(X[50000:50000 + n], Y[50000:50000 + n], p, 1e-6, chunk=3000)
    print('cfg2 slice: data rel %.2e kl rel %.2e' % (abs(eds - d_r) / abs(d_r), abs(kls - kl_r) / abs(kl_r)))
    assert abs(eds - d_r) <= 1e-8 * abs(d_r) and abs(kls - kl_r) <= 1e-9 * abs(kl_r)
    for k in ot.PARAM_KEYS:
        a, b = np.asarray(gs[k]).reshape(-1), np.asarray(g_r[k]).reshape(-1)
        assert np.max(np.abs(a - b)) <= 1e-6 * max(np.max(np.abs(b)), 1e-300), k     # tolerance 1e-6 relative, fp64
    out = engine.predict(p, X[:4000])
    import zigp_oracle as o
    ref = o.build_predict(X[:4000], p, 1e-6)
    for i in range(9):
        r = ref[i].reshape(-1)
        assert np.max(np.abs(out[i] - r)) <= 1e-7 * max(np.max(np.abs(r)), 1e-300), i


# ---- BASELINE.json configs[0], literally: toydata.mat, M = 50, Z as zero-inflated-gpflow.ipynb:107 --------------------------
def test_cfg1_toydata_M50_matches_oracle(engine):
    import scipy.io
    import zigp_oracle as o
    import zigp_oracle_torch as ot
    d = scipy.io.loadmat(os.path.join(ROOT, 'tests', 'golden', 'toydata.mat'))
    X, Y = d['x'].astype(np.float64), d['y'].astype(np.float64)
    assert X.shape == (450, 1)
    Z = np.delete(np.linspace(0, 10, 51, endpoint=False), 0)[:, None]      # ipynb:107 pattern with num_inducing = 51 -> M = 50
    assert Z.shape == (50, 1)
    ru = np.random.RandomState(5)
    p = dict(Zf=Z.copy(), Zg=Z.copy(), u_fm=0.01 * ru.randn(50, 1), u_gm=0.01 * ru.randn(50, 1),      # OnOffSVGP.py:56-57 scale
             u_fs_sqrt=np.ones((50, 1)), u_gs_sqrt=np.ones((50, 1)),                                   # :60-63
             ell_f=np.array([2.0]), ell_g=np.array([2.0]), var_f=1.0, var_g=5.0, noise=0.01)          # ipynb:99-104,134
    engine.set_chunk(32768)
    engine.set_data(X, Y)
    ed, kl, g = engine.elbo(p, jitter=1e-6)
    e_r, d_r, kl_r, g_r = ot.elbo_and_grad(X, Y, p, 1e-6)
    Kuu = o.rbf_K(Z, Z, p['ell_f'], 1.0) + 1e-6 * np.eye(50)
    cond = np.linalg.cond(Kuu)
    print('cfg1 literal: cond(Kuu) %.2e, elbo rel %.2e, kl rel %.2e' % (cond, abs((ed - kl) - e_r) / abs(e_r), abs(kl - kl_r) / abs(kl_r)))
    assert abs((ed - kl) - e_r) <= 1e-6 * abs(e_r)          # north-star tolerance: 1e-6 relative, fp64
    for k in ot.PARAM_KEYS:
        a, b = np.asarray(g[k]).reshape(-1), np.asarray(g_r[k]).reshape(-1)
        e = np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)
        print('  cfg1 grad %s relerr %.2e' % (k, e))
        assert e <= max(1e-6, 1e-13 * cond), (k, e)     # the rule of the other dense tests: gradients through K^-1 carry cond * eps
    out = engine.predict(p, X, jitter=1e-6)
    ref = o.build_predict(X, p, 1e-6)
    for i in range(9):
        r = ref[i].reshape(-1)
        assert np.max(np.abs(out[i] - r)) <= 1e-6 * max(np.max(np.abs(r)), 1e-300), i


def test_two_latents_in_one_launch_with_different_block_counts(engine):
    """The merged launches of the chunk loop (run_gemm2: A1, A2, J' of latent f and latent g as ONE grid each, paired tile order) with
    Mf = 300 (3 row blocks: a pair and a single middle tile per column panel) and Mg = 520 (5 row blocks): 304 column panels x (2 + 3)
    units = 1520 workgroups = 0.99 of three waves -> trmm_paired_pays.  Against the same step in 38 passes of 1024 rows, which takes the
    LPT order with one launch per latent (8 column panels x 5 units < 512), and against the oracle on a slice."""
    import zigp_oracle_torch as ot
    N = 38400
    X, Y, p = make_problem(N, 300, 3, seed=31, Mg=520, ell=0.3)
    engine.set_data(X, Y)
    engine.set_chunk(65536)
    assert engine.get_chunk_rows(520, N) == 38912               # one pass of 304 column panels: 1520 units
    ed, kl, g = engine.elbo(p, jitter=1e-6)
    engine.set_chunk(1024)
    ed2, kl2, g2 = engine.elbo(p, jitter=1e-6)
    print('  one pass vs 38: elbo_data %.2e' % (abs(ed - ed2) / abs(ed)), {k: '%.1e' % relerr(g[k], g2[k]) for k in g})
    assert abs(ed - ed2) <= 1e-9 * abs(ed) and kl == kl2      # (summation order over the points differs; u_m at scale 0.5, jitter 1e-6)
    for k in g:
        assert relerr(g[k], g2[k]) <= 1e-7, k
    engine.set_chunk(65536)
    # the single-stream schedule of zigp_set_overlap(0) keeps the point-wise stage in its own launch (with the overlap it rides in the
    # J' launch of both latents) and runs the side kernels on the main stream: only the order of independent work differs
    engine.set_overlap(False)
    ed0, kl0, g0 = engine.elbo(p, jitter=1e-6)
    engine.set_overlap(True)
    assert ed0 == ed and kl0 == kl and all(np.array_equal(np.asarray(g0[k]), np.asarray(g[k])) for k in g)
    rows = (12800, 12800 + 25600)                 # 200 panels x 5 units = 1000 workgroups = 0.98 of two waves: merged as well
    eds, _, gs = engine.elbo(p, jitter=1e-6, rows=rows, include_kl=False)
    sl = slice(*rows)
    e_r, d_r, kl_r, g_r = ot.elbo_and_grad(X[sl], Y[sl], p, 1e-6, include_kl=False)
    assert abs(eds - d_r) <= 1e-9 * abs(d_r)
    for k in ot.PARAM_KEYS:
        assert relerr(gs[k], np.asarray(g_r[k]).reshape(np.asarray(gs[k]).shape)) < 1e-6, k
