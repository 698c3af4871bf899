"""Full-size (BASELINE.json cfg3: N = 1e6, M = 1024, D = 3) properties that need no oracle at that size, plus an oracle
check on a slice.  Same synthetic generator as bench.py."""
import os
import sys

import numpy as np
import pytest

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
