"""GPU parity of the single-latent Kronecker heads (the reference's baselines) against the
LITERAL dense restatement of scripts/svgp.py:116-233 (Gaussian) and scripts/classifier.py:116-240 (Bernoulli / probit)."""
import numpy as np
import pytest

from conftest import relerr
from test_gpu_kron import make_kron_problem

pytestmark = pytest.mark.gpu

CASES = [(200, 6, 5), (1000, 10, 12), (700, 32, 32), (600, 10, 100)]   # the last one: the reference's [10, 100] grid (larger-grid kernels)


def _head_problem(N, M0, M1, lik, seed):
    X, Y, p = make_kron_problem(N, M0, M1, seed=seed)
    ph = {k: p[k] for k in ('Zf', 'ell_f', 'var_f', 'u_fm', 'u_fs_sqrt', 'noise')}
    if lik == 'bernoulli':
        Y = (Y > 0) * 1.0              # classifier.py:43-44
    return X, Y, ph


@pytest.mark.parametrize('lik', ['gaussian', 'bernoulli'])
@pytest.mark.parametrize('N,M0,M1', CASES)
def test_head_predict_matches_literal_oracle(engine, lik, N, M0, M1):
    import zigp_oracle as o
    X, Y, p = _head_problem(N, M0, M1, lik, seed=N + 7)
    for jit, f_mu in ((1e-5, 0.0), (1e-6, 0.3)):
        out = engine.kron_head_predict(p, X, lik, jitter=jit, f_mu=f_mu)
        ref = o.kron_head_predict(X, p, lik, jit, f_mu)
        if lik == 'gaussian':
            pairs = (('fmean', out[0], ref[0]), ('fvar', out[1], ref[1]), ('ymean', out[2], ref[0]), ('yvar', out[3], ref[1] + p['noise']))
        else:
            pairs = (('pfmean', out[2], ref[0]), ('pfvar', out[3], ref[1]), ('fmean', out[0], ref[2]), ('fvar', out[1], ref[3]))
        for name, a, b in pairs:
            e = relerr(a, np.asarray(b).reshape(-1))
            print('%s jitter %g %s relerr %.2e' % (lik, jit, name, e))
            assert e < 1e-6, (name, e)   # tolerance: 1e-6 relative, fp64 (BASELINE.json north_star)


@pytest.mark.parametrize('lik', ['gaussian', 'bernoulli'])
@pytest.mark.parametrize('N,M0,M1', CASES)
def test_head_elbo_and_gradient_match_literal_oracle(engine, lik, N, M0, M1):
    import zigp_oracle_torch as ot
    X, Y, p = _head_problem(N, M0, M1, lik, seed=N + 11)
    scale, f_mu = 105280.0 / N, (0.0 if lik == 'gaussian' else 0.2)
    ed, kl, g = engine.kron_head_elbo(p, X, Y, lik, jitter=1e-5, scale=scale, f_mu=f_mu)
    e_r, d_r, kl_r, g_r = ot.kron_head_elbo_and_grad(X, Y, p, lik, 1e-5, scale=scale, f_mu=f_mu)
    print('%s elbo %.10e ref %.10e  kl %.8e ref %.8e' % (lik, ed - kl, e_r, kl, kl_r))
    assert abs(ed - scale * d_r) <= 1e-7 * abs(scale * d_r)
    assert abs(kl - kl_r) <= 1e-7 * abs(kl_r)
    for k in ('Zf', 'ell_f', 'var_f'):
        for q in range(2):
            a, b = np.asarray(g[k][q]).reshape(-1), np.asarray(g_r[k][q]).reshape(-1)
            e = np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)
            print('  grad %s[%d] relerr %.2e (max |ref| %.3e)' % (k, q, e, np.max(np.abs(b))))
            assert e < 1e-6, (k, q, e)
    keys = ('u_fm', 'u_fs_sqrt', 'f_mu') + (('noise',) if lik == 'gaussian' else ())
    for k in keys:
        a, b = np.asarray(g[k]).reshape(-1), np.asarray(g_r[k]).reshape(-1)
        e = np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)
        print('  grad %s relerr %.2e' % (k, e))
        assert e < 1e-6, (k, e)
    if lik == 'bernoulli':
        assert g['noise'] == 0.0       # the probit head has no noise parameter (classifier.py declares none)


def test_head_rejects_onoff_and_bad_args(engine):
    X, Y, p = _head_problem(100, 4, 4, 'gaussian', seed=1)
    with pytest.raises(KeyError):
        engine.kron_head_elbo(p, X, Y, 'onoff')
    with pytest.raises(ValueError):
        engine.kron_head_elbo(p, X[:, :2], Y, 'gaussian')
    bad = dict(p, var_f=[np.array([-1.0]), np.array([1.0])])
    with pytest.raises(ValueError):
        engine.kron_head_elbo(bad, X, Y, 'gaussian')


def test_onoff_path_unchanged_after_head_calls(engine):
    """the heads share the context's Kronecker state with the two-latent path: interleaving must not leak state."""
    X, Y, p = make_kron_problem(400, 6, 7, seed=9)
    ed0, kl0, g0 = engine.kron_elbo(p, X, Y, jitter=1e-5, scale=2.0)
    ph = {k: p[k] for k in ('Zf', 'ell_f', 'var_f', 'u_fm', 'u_fs_sqrt', 'noise')}
    engine.kron_head_elbo(ph, X[:100], (Y[:100] > 0) * 1.0, 'bernoulli')
    engine.kron_head_predict(ph, X[:50], 'gaussian')
    ed1, kl1, g1 = engine.kron_elbo(p, X, Y, jitter=1e-5, scale=2.0)
    assert ed0 == ed1 and kl0 == kl1
    assert np.array_equal(g0['u_gm'], g1['u_gm']) and np.array_equal(g0['Zf'][0], g1['Zf'][0])
