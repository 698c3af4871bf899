"""GPU tests of the TF-stack look-alike (onofftf: onoff fit, predict_onoff) on the reference's precipitation data
(tests/golden/pptr.npz = data/pptr.pickle re-saved; time column / 1000 as scripts/create_cvsplits.py:17)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _pptr():
    d = np.load(os.path.join(GOLD, 'pptr.npz'))
    Xtr, Xte = d['Xtrain'].copy(), d['Xtest'].copy()
    Xtr[:, 2] /= 1000.0
    Xte[:, 2] /= 1000.0
    return Xtr, d['Ytrain'], Xte, d['Ytest']


def test_onoff_fit_and_predict_roundtrip(tmp_path, engine):
    from onofftf import onoff, predict_onoff
    Xtr, Ytr, Xte, Yte = _pptr()
    hist = []
    np.random.seed(0)
    out = onoff(Xtr, Ytr, Xte[:2000], Yte[:2000], str(tmp_path) + '/', num_iter=150, num_inducing_f=(10, 20), num_inducing_g=(10, 20),
                engine=engine, kmeans_seed=1, history=hist)
    assert set(out) == {'Xtrain', 'Ytrain', 'Xtest', 'Ytest', 'test_rmse', 'test_mae'}
    assert np.isfinite(out['test_rmse']) and np.isfinite(out['test_mae'])
    assert np.mean(hist[-20:]) < np.mean(hist[:20])            # the cost (-ELBO) goes down
    assert os.path.exists(str(tmp_path) + '/model.npz') and os.path.exists(str(tmp_path) + '/modelsumm.log')
    ptr, pte = predict_onoff(Xtr[:3000], Xte[:500], str(tmp_path) + '/', np.array([10, 20]), np.array([10, 20]), engine=engine)
    assert set(ptr) == {'gfmean', 'fmean', 'pgmean'} and pte['gfmean'].shape == (500, 1)
    assert np.all((pte['pgmean'] > 0) & (pte['pgmean'] < 1))


def test_full_batch_cfg5_step_matches_chunked_literal_oracle(engine):
    """cfg5: N = 105 280, 32 x 32 inducing grid, one full-batch ELBO value; oracle = literal dense order on 4 x 1000 rows."""
    import zigp_oracle as o
    from onofftf.model import init_params, engine_params
    Xtr, Ytr, _, _ = _pptr()
    np.random.seed(3)
    p = engine_params(init_params(Xtr, (32, 32), (32, 32), kmeans_seed=2))
    ed, kl, g = engine.kron_elbo(p, Xtr, Ytr, jitter=1e-5, scale=1.0)
    assert np.isfinite(ed) and np.isfinite(kl) and all(np.all(np.isfinite(np.asarray(v, dtype=float))) for v in (g['u_fm'], g['Zf'][0], g['Zg'][1]))
    idx = np.arange(0, 4000)
    ed_s, _, _ = engine.kron_elbo(p, Xtr[idx], Ytr[idx], jitter=1e-5, include_kl=False, need_grad=False)
    ref = sum(o.kron_elbo(Xtr[s:s + 1000], Ytr[s:s + 1000], p, 1e-5)[1] for s in range(0, 4000, 1000))
    print('cfg5 sample data term gpu %.10e oracle %.10e' % (ed_s, ref))
    assert abs(ed_s - ref) < 1e-6 * abs(ref)
