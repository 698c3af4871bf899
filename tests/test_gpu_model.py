"""GPU tests of the reference-facing model surface (onoffgpf look-alike) incl. the notebook recipe on toydata.mat."""
import os
import pickle

import numpy as np
import pytest
import scipy.io as sio

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _toy_model(num_inducing=10, seed=0):
    """zero-inflated-gpflow.ipynb:52-135 (data/toydata.mat is the reference's own data file)."""
    import onoffgpf
    from onoffgpf import OnOffSVGP, OnOffLikelihood
    mat = sio.loadmat(os.path.join(GOLD, 'toydata.mat'))
    X, Y = mat['x'], mat['y']
    kf = onoffgpf.kernels.RBF(1)
    kf.lengthscales = 2.
    kf.variance = 1.
    kg = onoffgpf.kernels.RBF(1)
    kg.lengthscales = 2.
    kg.variance = 5.
    Zf = np.delete(np.linspace(min(X), max(X), num_inducing, endpoint=False), 0).transpose().reshape(-1, 1)
    Zg = Zf.copy()
    np.random.seed(seed)      # the reference leaves the u_m init unseeded (OnOffSVGP.py:56-57)
    m = OnOffSVGP(X, Y, kernf=kf, kerng=kg, likelihood=OnOffLikelihood(), Zf=Zf, Zg=Zg)
    m.likelihood.variance = 0.01
    m.likelihood.variance.fixed = False
    return m, X, Y


def test_model_surface_matches_oracle_at_init():
    import zigp_oracle as o
    m, X, Y = _toy_model()
    p = m._values()
    e_r, d_r, klf, klg = o.elbo(X, Y, p, 1e-6)
    assert abs(m.compute_log_likelihood() - e_r) < 1e-8 * abs(e_r)
    assert abs(m.compute_prior_KL() - (klf + klg)) < 1e-9 * abs(klf + klg)
    out = m.predict_onoffgp(X)
    ref = o.build_predict(X, p, 1e-6)
    assert len(out) == 9 and all(a.shape == (X.shape[0], 1) for a in out)
    for a, b in zip(out, ref):
        assert np.max(np.abs(a - b)) <= 1e-8 * max(np.max(np.abs(b)), 1e-300)
    # attributes the plotter reads (onoffgpf/PlotOnOff1D.py:12-26)
    assert m.Xtrain.value.shape == X.shape and m.Zf.value.shape == (9, 1) and m.u_fs_sqrt.value.shape == (9, 1)
    assert m.likelihood.variance.value[0] == 0.01
    K = m.kernf.compute_K_symm(X)
    assert K.shape == (450, 450) and abs(K[0, 0] - 1.0) < 1e-12


def test_toy_fit_reaches_notebook_band(tmp_path):
    """Notebook: 8000 L-BFGS-B iterations -> ELBO 488.713 (zero-inflated-gpflow.ipynb:146), unseeded init,
    so only a loose band is meaningful."""
    m, X, Y = _toy_model()
    e0 = m.compute_log_likelihood()
    res = m.optimize(maxiter=8000)
    e1 = m.compute_log_likelihood()
    print('toy ELBO: init %.3f -> %.6f after %d its (%d evals); notebook 488.713' % (e0, e1, res.nit, res.nfev))
    assert e1 > e0
    assert 400.0 < e1 < 560.0
    # checkpoint round trip (savemodel, OnOffSVGP.py:154-158)
    f = m.savemodel(str(tmp_path / 'm.pickle'))
    m2 = pickle.load(open(f, 'rb'))
    assert abs(m2.compute_log_likelihood() - e1) < 1e-9 * abs(e1)


def test_minibatch_scale_and_adam():
    import onoffgpf
    from onoffgpf import OnOffSVGP, OnOffLikelihood
    mat = sio.loadmat(os.path.join(GOLD, 'toydata.mat'))
    X, Y = mat['x'], mat['y']
    Z = np.linspace(1, 9, 9)[:, None]
    np.random.seed(1)
    m = OnOffSVGP(X, Y, onoffgpf.kernels.RBF(1, lengthscales=2.), onoffgpf.kernels.RBF(1, lengthscales=2., variance=5.),
                  OnOffLikelihood(), Z, Z.copy(), minibatch_size=100)
    e = [m.compute_log_likelihood() for _ in range(3)]
    assert len(set(e)) == 3                      # different minibatches
    m.optimize(method='adam', maxiter=50, learning_rate=0.01)
    assert np.isfinite(m.compute_log_likelihood())


def test_mean_function_model_surface(engine):
    """OnOffSVGP(mean_function=Constant/Linear): trainable through optimize(), shifts fmean in predict_onoffgp (OnOffSVGP.py:134)."""
    import onoffgpf
    from onoffgpf import OnOffSVGP, OnOffLikelihood
    from onoffgpf.mean_functions import Constant, Linear, Zero
    rs = np.random.RandomState(0)
    X = rs.rand(400, 1) * 10
    Y = np.where(np.sin(X) > 0, 3.0 + 0.5 * X + 0.1 * rs.randn(400, 1), 0.0)
    Z = np.linspace(0.5, 9.5, 12)[:, None]

    def mk(mf):
        np.random.seed(1)
        return OnOffSVGP(X, Y, kernf=onoffgpf.kernels.RBF(1, lengthscales=2.0), kerng=onoffgpf.kernels.RBF(1, lengthscales=2.0, variance=5.0),
                         likelihood=OnOffLikelihood(), Zf=Z, Zg=Z.copy(), mean_function=mf)
    m0, mc, ml = mk(None), mk(Constant(2.0)), mk(Linear(np.array([[0.5]]), 3.0))
    assert isinstance(m0.mean_function, Zero)
    f0, fc, fl = m0.predict_onoffgp(X)[3], mc.predict_onoffgp(X)[3], ml.predict_onoffgp(X)[3]
    assert np.allclose(fc - f0, 2.0, atol=1e-10) and np.allclose(fl - f0, 3.0 + 0.5 * X, atol=1e-10)
    assert np.allclose(ml.mean_function(X), 3.0 + 0.5 * X)
    e_before = mc.compute_log_likelihood()
    c_before = float(mc.mean_function.c.value[0])
    mc.optimize(maxiter=25)
    assert mc.compute_log_likelihood() > e_before and float(mc.mean_function.c.value[0]) != c_before
    ml.optimize(maxiter=25)
    assert ml.mean_function.A.value.shape == (1, 1) and np.isfinite(ml.compute_log_likelihood())
    with pytest.raises(TypeError):
        mk(lambda x: 0 * x)
