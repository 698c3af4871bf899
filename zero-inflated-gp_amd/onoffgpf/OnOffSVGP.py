"""OnOffSVGP look-alike (onoffgpf/OnOffSVGP.py:18-204) driving the MI355X engine through the C-ABI.

Same constructor signature, attributes and methods the reference exposes / its notebook and plotter use:
  OnOffSVGP(X, Y, kernf, kerng, likelihood, Zf, Zg, mean_function=None, minibatch_size=None, name='model')
  .optimize(maxiter=...)  .compute_log_likelihood()  .predict_onoffgp(Xnew)  .compute_prior_KL()  .savemodel(fname)
  .Xtrain .Ytrain .Zf .Zg .u_fm .u_gm .u_fs_sqrt .u_gs_sqrt .kernf .kerng .likelihood.variance
whiten=False and q_diag=True are hard-coded in the reference (:33-34).  mean_function (:29,134): onoffgpf.mean_functions.Zero
(default), Constant or Linear -- evaluated, and differentiated, inside the engine's point-wise kernel.
"""
import pickle
import time
from collections import OrderedDict

import numpy as np

import zigp
from zigp.optim import ParamSet, lbfgsb, AdamGroups
from zigp.transforms import positive
from .param import Param, DataHolder, Parameterized
from .mean_functions import MeanFunction, Zero

JITTER = 1e-6   # gpflow settings.numerics.jitter_level default (OnOffSVGP.py:96-97) [GPflow-recall]


class OnOffSVGP(Parameterized):
    def __init__(self, X, Y, kernf, kerng, likelihood, Zf, Zg, mean_function=None, minibatch_size=None, name='model',
                 device=0):
        self.mean_function = mean_function or Zero()                 # :29
        if not isinstance(self.mean_function, MeanFunction):
            raise TypeError('mean_function must be an onoffgpf.mean_functions.{Zero, Constant, Linear}')
        X, Y = np.asarray(X, dtype=np.float64), np.asarray(Y, dtype=np.float64)
        if Y.ndim != 2 or Y.shape[1] != 1:
            raise ValueError('Y must be (N,1): num_latent is 1 (OnOffSVGP.py:45)')
        self.name = name
        self.kernf, self.kerng, self.likelihood = kernf, kerng, likelihood
        self.whiten, self.q_diag = False, True                       # :33-34
        self.Xtrain, self.Ytrain = DataHolder(X), DataHolder(Y)      # :37-39
        self.num_data = X.shape[0]
        self.num_latent = Y.shape[1]
        self.minibatch_size = self.num_data if minibatch_size is None else int(minibatch_size)   # :42-43
        self._rng = np.random.RandomState(0)                         # :46-47 (same seed for X and Y)
        self.Zf, self.Zg = Param(np.array(Zf, dtype=np.float64)), Param(np.array(Zg, dtype=np.float64))   # :50-51
        self.num_inducing_f, self.num_inducing_g = self.Zf.value.shape[0], self.Zg.value.shape[0]
        self.u_fm = Param(np.random.randn(self.num_inducing_f, self.num_latent) * 0.01)   # :56 (unseeded, as the reference)
        self.u_gm = Param(np.random.randn(self.num_inducing_g, self.num_latent) * 0.01)   # :57
        self.u_fs_sqrt = Param(np.ones((self.num_inducing_f, self.num_latent)), positive)  # :60-61
        self.u_gs_sqrt = Param(np.ones((self.num_inducing_g, self.num_latent)), positive)  # :62-63
        self._device = int(device)
        self._engine = zigp.reference_engine(self._device)  # raises if libzigp.so / GPU is missing: no CPU fallback; tf.cholesky's pivot rule
        self._resident = False

    # ---- parameter plumbing -----------------------------------------------------------------
    def _pset(self):
        return ParamSet(OrderedDict([
            ('Zf', self.Zf), ('Zg', self.Zg), ('u_fm', self.u_fm), ('u_gm', self.u_gm),
            ('u_fs_sqrt', self.u_fs_sqrt), ('u_gs_sqrt', self.u_gs_sqrt),
            ('ell_f', self.kernf.lengthscales), ('ell_g', self.kerng.lengthscales),
            ('var_f', self.kernf.variance), ('var_g', self.kerng.variance), ('noise', self.likelihood.variance)]
            + list(self.mean_function.trainables().items())))

    def _values(self):
        a, b = self.mean_function.linear_form(self.Xtrain.value.shape[1])
        mf = {k: v for k, v in (('mean_a', a), ('mean_b', b)) if v is not None}
        return dict(mf, Zf=self.Zf.value, Zg=self.Zg.value, u_fm=self.u_fm.value, u_gm=self.u_gm.value,
                    u_fs_sqrt=self.u_fs_sqrt.value, u_gs_sqrt=self.u_gs_sqrt.value,
                    ell_f=self.kernf.ell_vector(), ell_g=self.kerng.ell_vector(),
                    var_f=float(self.kernf.variance.value.reshape(-1)[0]), var_g=float(self.kerng.variance.value.reshape(-1)[0]),
                    noise=float(self.likelihood.variance.value.reshape(-1)[0]))

    def _fold_grads(self, g):
        """ARD engine gradient -> the shape of the Param (a scalar lengthscale sums its D copies)."""
        out = dict(g)
        for k, kern in (('ell_f', self.kernf), ('ell_g', self.kerng)):
            if kern.lengthscales.value.size == 1:
                out[k] = np.array([np.sum(g[k])])
        for k in ('var_f', 'var_g', 'noise'):
            out[k] = np.array([g[k]])
        if 'mean_b' in g:
            out['mean_b'] = np.array([g['mean_b']])
        return out

    def _load_batch(self):
        """X and Y go to HBM once; a minibatch (MinibatchData, :46-47) is a row-index sample gathered on the device per step."""
        if not self._resident:
            self._engine.set_data(self.Xtrain.value, self.Ytrain.value)
            self._resident = True
        if self.minibatch_size >= self.num_data:
            self._engine.select_rows(None)       # back to the full resident set (a minibatch_size raised after a minibatch step must not leave its last sample active)
            return 1.0
        # GPflow 0.4 MinibatchData picks its index manager by the batch fraction [GPflow-recall; not in the reference tree, unverified]:
        # up to one half sampling WITH replacement (rng.randint), ABOVE one half a fresh permutation's head (without replacement) -- the
        # boundary case of exactly one half goes with randint, as the `fraction > 0.5` test of that recollection says
        if 2 * self.minibatch_size <= self.num_data:
            idx = self._rng.randint(self.num_data, size=self.minibatch_size)
        else:
            idx = self._rng.permutation(self.num_data)[:self.minibatch_size]
        self._engine.select_rows(idx)
        return float(self.num_data) / float(self.minibatch_size)          # :119-120

    def _elbo(self, need_grad):
        scale = self._load_batch()
        ed, kl, g = self._engine.elbo(self._values(), jitter=JITTER, scale=scale, need_grad=need_grad)
        return ed - kl, (self._fold_grads(g) if need_grad else None)

    # ---- reference surface ------------------------------------------------------------------
    def compute_log_likelihood(self):
        """build_likelihood value (OnOffSVGP.py:107-122)."""
        return self._elbo(False)[0]

    def compute_prior_KL(self):
        """build_prior_KL (OnOffSVGP.py:73-105,164-166)."""
        return float(np.sum(self._engine.prior_kl(self._values(), jitter=JITTER)))

    def predict_onoffgp(self, Xnew):
        """build_predict (OnOffSVGP.py:124-152,160-162): 9 arrays of shape (N,1), order of :152."""
        out = self._engine.predict(self._values(), np.asarray(Xnew, dtype=np.float64), jitter=JITTER)
        return tuple(out[i].reshape(-1, 1) for i in range(9))

    def optimize(self, method='L-BFGS-B', maxiter=1000, disp=False, callback=None, learning_rate=0.01, **kw):
        """GPflow Model.optimize: scipy L-BFGS-B on the free state (default), or Adam when method='adam'
        (the commented alternative at zero-inflated-gpflow.ipynb:155)."""
        pset = self._pset()

        def vg(_values):
            return self._elbo(True)

        if str(method).lower() in ('l-bfgs-b', 'lbfgsb'):
            return lbfgsb(pset, vg, maxiter=maxiter, disp=disp, callback=callback, **kw)
        if str(method).lower() == 'adam':
            for p in pset.params.values():
                p.learning_rate = learning_rate
            opt = AdamGroups(pset)
            for it in range(maxiter):
                elbo, g = self._elbo(True)
                opt.step(g)
                if callback is not None:
                    callback(it, elbo)
            return None
        raise ValueError('unknown method %r' % (method,))

    def savemodel(self, fname=None):
        """pickle.dump of the model (OnOffSVGP.py:154-158); the engine handle is dropped and re-made on load."""
        if fname is None:
            fname = 'pm_' + time.strftime('%Y%m%d-%H%M') + '_' + str(self.name) + '.pickle'
        with open(fname, 'wb') as f:
            pickle.dump(self, f)
        return fname

    def __getstate__(self):
        d = dict(self.__dict__)
        d['_engine'] = None
        d['_resident'] = False
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self.__dict__.setdefault('mean_function', Zero())
        self.__dict__['_engine'] = zigp.reference_engine(self.__dict__.setdefault('_device', 0))   # the device it was fitted on

    @staticmethod
    def ProbitExpectations(gmean, gvar):
        """Host (NumPy) evaluation of OnOffSVGP.py:168-204 for inspection; the engine fuses the same formulas."""
        from scipy.special import erf
        z = gmean / np.sqrt(1. + gvar)
        a = 1 / np.sqrt(1. + (2 * gvar))
        cdfz = 0.5 * (1.0 + erf(z / np.sqrt(2.0))) * (1. - 2.e-3) + 1.e-3
        tz = np.arctan(a) / (2 * np.pi) * np.exp(-0.5 * np.square(z) * (np.square(a) + 1))
        pgmeansq = cdfz - 2. * tz
        pgvar = cdfz - 2. * tz - np.square(cdfz)
        return cdfz, (pgmeansq + np.abs(pgmeansq)) / 2., (pgvar + np.abs(pgvar)) / 2.
