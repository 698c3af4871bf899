"""GPU results against the COMMITTED golden fixtures alone (tests/golden/*.npz) -- no oracle code runs here: the engine must
reproduce the frozen numbers (G2: dense path on two seeded problems, value / 9-tuple / KL / full gradient, scale 1.3;
G4: Kronecker on/off and the Gaussian / Bernoulli heads on a seeded minibatch).  Tolerance 1e-6 relative, fp64 (north star)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')
PKEYS = ('Zf', 'Zg', 'u_fm', 'u_gm', 'u_fs_sqrt', 'u_gs_sqrt', 'ell_f', 'ell_g', 'var_f', 'var_g', 'noise')


def _rel(a, b):
    a, b = np.asarray(a, dtype=float).reshape(-1), np.asarray(b, dtype=float).reshape(-1)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.mark.parametrize('tag', ['toy', 'd3'])
def test_dense_matches_g2_fixture(engine, tag):
    g = np.load(os.path.join(GOLD, 'g2_dense_oracle.npz'))
    X, Y = g[tag + '_X'], g[tag + '_Y']
    p = {k: g['%s_p_%s' % (tag, k)] for k in PKEYS}
    engine.set_chunk(32768)
    engine.set_data(X, Y)
    ed, kl, gr = engine.elbo(p, jitter=1e-6, scale=1.3)
    assert abs((ed - kl) - float(g[tag + '_elbo'])) <= 1e-6 * abs(float(g[tag + '_elbo']))
    assert abs(ed - 1.3 * float(g[tag + '_data'])) <= 1e-6 * abs(1.3 * float(g[tag + '_data']))
    assert abs(kl - float(g[tag + '_klf']) - float(g[tag + '_klg'])) <= 1e-6 * abs(kl)
    klfg = engine.prior_kl(p, jitter=1e-6)
    assert abs(klfg[0] - float(g[tag + '_klf'])) <= 1e-6 * abs(klfg[0]) and abs(klfg[1] - float(g[tag + '_klg'])) <= 1e-6 * abs(klfg[1])
    out = engine.predict(p, X, jitter=1e-6)
    for i in range(9):
        assert _rel(out[i], g[tag + '_pred'][i]) <= 1e-6, i
    for k in PKEYS:
        e = _rel(gr[k], g['%s_g_%s' % (tag, k)])
        print('%s grad %s rel %.2e' % (tag, k, e))
        assert e <= 1e-6, k


def test_kronecker_matches_g4_fixture(engine):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
    from make_golden import kron_problem            # the seeded inputs only; no oracle arithmetic
    g = np.load(os.path.join(GOLD, 'g4_kron_oracle.npz'))
    X, Y, p = kron_problem()
    ed, kl, _ = engine.kron_elbo(p, X, Y, jitter=1e-5, scale=4.0)
    assert abs((ed - kl) - float(g['onoff_elbo'])) <= 1e-6 * abs(float(g['onoff_elbo']))
    assert abs(kl - float(g['onoff_klf']) - float(g['onoff_klg'])) <= 1e-6 * abs(kl)
    out = engine.kron_predict(p, X, jitter=1e-6, g_offset=-1.0)
    for i in range(9):
        assert _rel(out[i], g['onoff_pred'][i]) <= 1e-6, i
    ph = {k: p[k] for k in ('Zf', 'ell_f', 'var_f', 'u_fm', 'u_fs_sqrt', 'noise')}
    for lik, Yl, rows in (('gaussian', Y, (0, 1)), ('bernoulli', (Y > 0) * 1.0, (2, 3, 0, 1))):
        ed, kl, _ = engine.kron_head_elbo(ph, X, Yl, lik, jitter=1e-5, scale=4.0, f_mu=0.2)
        assert abs((ed - kl) - float(g[lik + '_elbo'])) <= 1e-6 * abs(float(g[lik + '_elbo']))
        assert abs(kl - float(g[lik + '_kl'])) <= 1e-6 * abs(kl)
        out = engine.kron_head_predict(ph, X, lik, jitter=1e-6, f_mu=0.2)
        for j, r in enumerate(rows):               # fixture row order: oracle's return order; engine rows: fmean, fvar, pfmean, pfvar
            assert _rel(out[r], g[lik + '_pred'][j]) <= 1e-6, (lik, j)
