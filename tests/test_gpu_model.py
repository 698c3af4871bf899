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
    assert 0.95 * 488.713 < e1 < 1.05 * 488.713          # +-5 % of the notebook's number (seed 0: 480.5; see the 8-seed test)
    # checkpoint round trip (savemodel, OnOffSVGP.py:154-158)
    f = m.savemodel(str(tmp_path / 'm.pickle'))
    m2 = pickle.load(open(f, 'rb'))
    assert abs(m2.compute_log_likelihood() - e1) < 1e-9 * abs(e1)


def test_toy_fit_eight_seeds_bracket_the_notebook_number():
    """The notebook's 488.7130771963765 (zero-inflated-gpflow.ipynb:146) is ONE draw of an unseeded init (OnOffSVGP.py:56-57) read
    off after 8000 L-BFGS-B iterations -- BEFORE convergence: every seed is still climbing at 8000 (scipy stops on the iteration
    limit) and plateaus 20-50 units higher (tools/toy_long.py: 478 ... 509).  Evidence that the recipe reproduces it: the values at
    8000 iterations from 8 seeds reach to within 1 % below it, none is further than 8 % away, and the best seed continued to
    convergence passes it -- the notebook number is bracketed by [value at 8000 its, converged value] of the same recipe.
    The distribution goes to gpurun_out/toy_seeds.json (quoted in DESIGN.md)."""
    import json
    NOTEBOOK = 488.7130771963765
    runs, models = [], []
    for seed in range(8):
        m, X, Y = _toy_model(seed=seed)
        res = m.optimize(maxiter=8000)
        runs.append(dict(seed=seed, elbo=float(m.compute_log_likelihood()), nit=int(res.nit), nfev=int(res.nfev)))
        models.append(m)
        print('toy seed %d: ELBO %.6f after %d its (%d evals)' % (seed, runs[-1]['elbo'], res.nit, res.nfev))
    e = np.array([r['elbo'] for r in runs])
    print('toy ELBO over 8 seeds at 8000 its: min %.4f median %.4f max %.4f (notebook 488.7131)' % (e.min(), np.median(e), e.max()))
    best = int(np.argmax(e))
    res = models[best].optimize(maxiter=40000)
    conv = float(models[best].compute_log_likelihood())
    print('best seed %d continued: ELBO %.6f after %d more its' % (best, conv, res.nit))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        json.dump(dict(notebook=NOTEBOOK, runs=runs, best_seed=best, best_seed_converged=conv, extra_its=int(res.nit)),
                  open(os.path.join(out, 'toy_seeds.json'), 'w'), indent=1)
    except OSError:
        pass
    assert e.max() >= 0.99 * NOTEBOOK                      # within 1 % at the notebook's own iteration count
    assert np.all(e >= 0.92 * NOTEBOOK) and np.all(e <= 1.05 * NOTEBOOK)
    assert e.max() <= NOTEBOOK + 25.0 and conv >= NOTEBOOK  # [value at 8000 its, converged value] brackets the notebook number


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
    # the minibatch estimate (row sample gathered on the device, zigp_select_rows; scaled by num_data / batch, OnOffSVGP.py:119-120) is an
    # unbiased estimate of the full-batch bound: 200 draws average to it within a few standard errors
    np.random.seed(1)
    full = OnOffSVGP(X, Y, onoffgpf.kernels.RBF(1, lengthscales=2.), onoffgpf.kernels.RBF(1, lengthscales=2., variance=5.),
                     OnOffLikelihood(), Z, Z.copy())
    ref = full.compute_log_likelihood()
    draws = np.array([m.compute_log_likelihood() for _ in range(200)])
    se = draws.std() / np.sqrt(len(draws))
    print('full %.3f, minibatch mean %.3f +- %.3f' % (ref, draws.mean(), se))
    assert abs(draws.mean() - ref) < 5 * se
    # from half the data up the sample is the head of a fresh permutation: no repeated rows
    m2 = OnOffSVGP(X, Y, onoffgpf.kernels.RBF(1, lengthscales=2.), onoffgpf.kernels.RBF(1, lengthscales=2., variance=5.),
                   OnOffLikelihood(), Z, Z.copy(), minibatch_size=300)
    seen = []
    orig = m2._engine.select_rows
    m2._engine.select_rows = lambda idx: (seen.append(np.array(idx)), orig(idx))[1]
    m2.compute_log_likelihood()
    assert len(seen) == 1 and len(np.unique(seen[0])) == 300
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


def test_look_alikes_run_on_the_reference_cholesky_rule(tmp_path, monkeypatch):
    """Results identical to the reference on the same inputs includes where it FAILS: tf.cholesky (onofftf/main.py:200,268,355) rejects a
    non-positive pivot only, so the engines the look-alikes create (zigp.reference_engine: OnOffSVGP, onoff(), predict_onoff(), the
    likelihood heads, kernels.RBF.compute_K) run with zigp_set_pivot_rtol(ctx, 0); the bare engine keeps its stricter 8 eps (variance +
    jitter) guard as an option.  Two inducing points 2e-8 lengthscales apart, jitter 0: the second pivot is 1 - exp(-4e-16) ~ 4e-16 -- positive,
    below 8 eps: accepted by the model's engine, ZIGP_ENOTPD (pivot index 2) on a bare one.  Exact duplicates (pivot 0) fail under both."""
    import zigp
    from zigp.engine import reference_engine
    m, X, Y = _toy_model()
    Z = np.array([[0.0], [2e-8], [1.0], [2.5], [4.0], [7.0]])
    p = dict(Zf=Z, Zg=Z + 0.25, u_fm=np.zeros((6, 1)), u_gm=np.zeros((6, 1)), u_fs_sqrt=np.ones((6, 1)), u_gs_sqrt=np.ones((6, 1)),
             ell_f=np.array([1.0]), ell_g=np.array([1.0]), var_f=1.0, var_g=1.0, noise=0.01)
    eng = m._engine
    eng.set_data(X, Y.reshape(-1))
    ed, kl, _ = eng.elbo(p, jitter=0.0, need_grad=False)             # accepted, as tf.cholesky accepts it
    assert np.isfinite(ed) and np.isfinite(kl)
    bare = zigp.DenseEngine(0)
    try:
        bare.set_data(X, Y.reshape(-1))
        with pytest.raises(zigp.NotPositiveDefiniteError):
            bare.elbo(p, jitter=0.0, need_grad=False)
        assert bare.lib.zigp_last_info(bare.ctx) == 2
        bare.set_pivot_rtol(0.0)                                      # ... and the same engine under the reference's rule
        assert bare.elbo(p, jitter=0.0, need_grad=False)[0] == ed
    finally:
        bare.close()
    dup = dict(p, Zf=np.array([[0.0], [0.0], [1.0], [2.5], [4.0], [7.0]]))
    with pytest.raises(zigp.NotPositiveDefiniteError):                # pivot exactly 0: "not positive definite" in the reference too
        eng.elbo(dup, jitter=0.0, need_grad=False)
    assert eng.lib.zigp_last_info(eng.ctx) == 2
    m._resident = False                                               # the model's own data goes back in on its next call
    # every look-alike that builds its own engine takes it from zigp.reference_engine
    made = []
    monkeypatch.setattr(zigp, 'reference_engine', lambda device=0: made.append(device) or reference_engine(device))
    from onofftf import onoff, predict_onoff
    d = np.load(os.path.join(GOLD, 'pptr.npz'))
    Xtr, Ytr, Xte, Yte = d['Xtrain'][:3000].copy(), d['Ytrain'][:3000], d['Xtest'][:200].copy(), d['Ytest'][:200]
    Xtr[:, 2] /= 1000.0
    Xte[:, 2] /= 1000.0
    onoff(Xtr, Ytr, Xte, Yte, str(tmp_path) + '/', num_iter=4, num_inducing_f=(6, 8), num_inducing_g=(6, 8), num_minibatch=500,
          log_every=2, save_every=4, kmeans_seed=1)
    predict_onoff(Xtr[:500], None, str(tmp_path) + '/', np.array([6, 8]), np.array([6, 8]))
    m2, _, _ = _toy_model()
    assert len(made) == 3 and m2._engine is not m._engine
