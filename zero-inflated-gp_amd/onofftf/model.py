"""Parameter set of the Kronecker zero-inflated GP exactly as scripts/onoff.py:51-137 declares it
(names follow the TF variable scopes f_kern/, g_kern/, likelihood/, f_ind/, g_ind/), + npz checkpoints in place of
tf.train.Saver (onofftf/utils.py:61-73)."""
from collections import OrderedDict

import numpy as np

from zigp.optim import ParamSet
from zigp.transforms import Log1pe, positive
from .main import Param


def init_params(Xtrain, num_inducing_f, num_inducing_g, init_noisevar=0.01, kern_lr=1e-3, indp_lr=1e-3, rng=None, kmeans_seed=None):
    """scripts/onoff.py:51-137 (fit, noise 0.01) / onofftf/onoffpred.py:21-105 (predict, noise 0.001)."""
    from scipy.cluster.vq import kmeans
    rng = rng or np.random
    init_ell = [np.array([8., 8.]), np.array([5. / 1000])]                     # :57,60
    Zs = kmeans(Xtrain[:, 0:2], int(num_inducing_f[0]), seed=kmeans_seed)[0]      # :67 (unseeded in the reference)
    if Zs.shape[0] < int(num_inducing_f[0]):                                     # kmeans may return fewer centroids
        extra = Xtrain[rng.choice(Xtrain.shape[0], int(num_inducing_f[0]) - Zs.shape[0], replace=False), 0:2]
        Zs = np.vstack([Zs, extra + 1e-3])
    Zt = np.linspace(Xtrain[:, 2].min(), Xtrain[:, 2].max(), int(num_inducing_f[1]))[:, None]   # :68
    Zs_g = Zs if int(num_inducing_g[0]) == Zs.shape[0] else kmeans(Xtrain[:, 0:2], int(num_inducing_g[0]), seed=kmeans_seed)[0]
    Zt_g = Zt if int(num_inducing_g[1]) == Zt.shape[0] else np.linspace(Xtrain[:, 2].min(), Xtrain[:, 2].max(), int(num_inducing_g[1]))[:, None]
    Mf, Mg = int(np.prod(num_inducing_f)), int(np.prod(num_inducing_g))
    p = OrderedDict()
    for i in range(2):
        p['f_kern/lengthscale_%d' % i] = Param(init_ell[i], Log1pe(), name='lengthscale', learning_rate=kern_lr)
        p['f_kern/variance_%d' % i] = Param([20.], Log1pe(), name='variance', learning_rate=kern_lr)      # :58
        p['g_kern/lengthscale_%d' % i] = Param(init_ell[i], Log1pe(), name='lengthscale', learning_rate=kern_lr)
        p['g_kern/variance_%d' % i] = Param([10.], Log1pe(), name='variance', learning_rate=kern_lr)      # :61
    p['likelihood/variance'] = Param(init_noisevar, Log1pe(), name='variance', learning_rate=kern_lr)     # :63,102-104
    p['f_ind/z_0'], p['f_ind/z_1'] = Param(Zs.copy(), name='z', learning_rate=indp_lr), Param(Zt.copy(), name='z', learning_rate=indp_lr)
    p['f_ind/value'] = Param(rng.randn(Mf, 1) * 0.1, name='value', learning_rate=indp_lr)                 # :71
    p['f_ind/variance'] = Param(np.ones((Mf, 1)), positive, name='variance', learning_rate=indp_lr)       # :72,113-115
    p['g_ind/z_0'], p['g_ind/z_1'] = Param(Zs_g.copy(), name='z', learning_rate=indp_lr), Param(Zt_g.copy(), name='z', learning_rate=indp_lr)
    p['g_ind/value'] = Param(rng.randn(Mg, 1) * 0.1, name='value', learning_rate=indp_lr)                 # :75
    p['g_ind/variance'] = Param(np.ones((Mg, 1)), positive, name='variance', learning_rate=indp_lr)
    return ParamSet(p)


def engine_params(pset):
    v = {k: q.value for k, q in pset.params.items()}
    return dict(Zf=[v['f_ind/z_0'], v['f_ind/z_1']], Zg=[v['g_ind/z_0'], v['g_ind/z_1']],
                ell_f=[v['f_kern/lengthscale_0'], v['f_kern/lengthscale_1']], ell_g=[v['g_kern/lengthscale_0'], v['g_kern/lengthscale_1']],
                var_f=[v['f_kern/variance_0'], v['f_kern/variance_1']], var_g=[v['g_kern/variance_0'], v['g_kern/variance_1']],
                u_fm=v['f_ind/value'], u_gm=v['g_ind/value'], u_fs_sqrt=v['f_ind/variance'], u_gs_sqrt=v['g_ind/variance'],
                noise=v['likelihood/variance'])


def named_grads(g):
    """engine gradient dict -> the Param names above."""
    out = {'likelihood/variance': np.array([g['noise']])}
    for tag in ('f', 'g'):
        for i in range(2):
            out['%s_kern/lengthscale_%d' % (tag, i)] = np.asarray(g['ell_' + tag][i])
            out['%s_kern/variance_%d' % (tag, i)] = np.array([g['var_' + tag][i]])
            out['%s_ind/z_%d' % (tag, i)] = np.asarray(g['Z' + tag][i])
        out['%s_ind/value' % tag] = np.asarray(g['u_%sm' % tag])
        out['%s_ind/variance' % tag] = np.asarray(g['u_%ss_sqrt' % tag])
    return out


# block order of zigp_kron_fit_steps' free-state vector (include/zigp.h): per latent Z0, Z1, u, s, ell0, ell1, var0, var1; then the noise
FIT_BLOCK_NAMES = tuple('%s_%s' % (tag, nm) for tag in ('f', 'g')
                        for nm in ('ind/z_0', 'ind/z_1', 'ind/value', 'ind/variance', 'kern/lengthscale_0', 'kern/lengthscale_1',
                                   'kern/variance_0', 'kern/variance_1')) + ('likelihood/variance',)


class KronDeviceFit:
    """The Adam state of the Kronecker on/off fit in the layout of zigp_kron_fit_steps, for a ParamSet made by init_params: the flat free
    state x and the moments m, v live here between calls (the engine updates them in place), `steps` advances them on the device and
    writes the constrained values back into the ParamSet.  zigp.optim.AdamGroups on the same ParamSet is the host-side checker."""

    def __init__(self, engine, pset, beta1=0.9, beta2=0.999, eps=1e-8):
        from zigp.transforms import Log1pe
        self.engine, self.pset = engine, pset
        self.beta1, self.beta2, self.eps = beta1, beta2, eps
        ps = [pset.params[k] for k in FIT_BLOCK_NAMES]
        if any(q.fixed for q in ps):
            raise ValueError('the device fit loop trains every parameter (fixed parameters: use the host loop)')
        for q in ps:
            if not isinstance(q.transform, Log1pe) and type(q.transform).__name__ != 'Identity':
                raise ValueError('unsupported transform %r' % (q.transform,))
            if isinstance(q.transform, Log1pe) and q.transform._lower != 1e-6:
                raise ValueError('the device fit loop implements Log1pe with lower = 1e-6')
        self.positive = [isinstance(q.transform, Log1pe) for q in ps]
        self.lr = [float(q.learning_rate) for q in ps]
        self.sizes = [q.value.size for q in ps]
        self.x = np.concatenate([q.free() for q in ps])
        self.m, self.v = np.zeros_like(self.x), np.zeros_like(self.x)
        self.t = 0
        self._written = [q.value.copy() for q in ps]      # what the ParamSet held when x was last derived from / written to it
        v = pset.params
        self.shape = dict(M0f=v['f_ind/z_0'].value.shape[0], M1f=v['f_ind/z_1'].value.shape[0], M0g=v['g_ind/z_0'].value.shape[0],
                          M1g=v['g_ind/z_1'].value.shape[0], D0=v['f_ind/z_0'].value.shape[1], D1=v['f_ind/z_1'].value.shape[1])

    def steps(self, row_begin, batch, jitter, scale, Xw=None, Yw=None, include_kl=True):
        """len(row_begin) iterations on the resident data set (engine.set_data): returns (elbo_data, kl) per step; the ParamSet holds the
        constrained values after the last one.  If the engine raises in step k (a Cholesky failure), x / m / v are the state after the k
        updates that WERE applied, self.t has advanced by k, and the exception carries `steps_applied`, `elbo_data`, `kl` of those steps:
        a caller that catches it and goes on (with more jitter, say) continues with the right iteration count and bias correction.
        The ParamSet is read again when something other than this object changed it since the last call (load_checkpoint, an assignment
        to .value): see resync().  include_kl: False on the ranks > 0 of a data-parallel fit (zigp.parallel.ShardedKronFit)."""
        if self._stale():
            self.resync()
        try:
            out = self.engine.kron_fit_steps(self.shape, self.x, self.m, self.v, self.lr, self.positive, self.t, row_begin, batch, jitter=jitter,
                                             scale=scale, Xw=Xw, Yw=Yw, beta1=self.beta1, beta2=self.beta2, eps=self.eps, include_kl=include_kl)
            self.t += len(row_begin)
        except Exception as e:
            self.t += int(getattr(e, 'steps_applied', 0))
            raise
        finally:
            self.sync_params()
        return out

    def _stale(self):
        """did anyone else write the ParamSet since sync_params? (cheap: ~20 small arrays against the copies kept there)"""
        # equal_nan: a parameter that HAS gone NaN (a diverged step) is still the value this object wrote -- NaN != NaN must not turn every
        # later call into a resync from free(NaN)
        return any(not np.array_equal(self.pset.params[k].value, w, equal_nan=True) for k, w in zip(FIT_BLOCK_NAMES, self._written))

    def resync(self, reset=False):
        """Take the free state from the ParamSet again (after load_checkpoint or a manual assignment).  Adam's moments and the iteration
        count are kept -- right for a nudged or restored-and-continued fit; reset=True zeroes them (t = 0), for parameters that are
        unrelated to the ones trained so far (load_checkpoint(..., fitter=...) of another run)."""
        self.x = np.concatenate([self.pset.params[k].free() for k in FIT_BLOCK_NAMES])
        if reset:
            self.m[:] = 0.0
            self.v[:] = 0.0
            self.t = 0
        self._written = [self.pset.params[k].value.copy() for k in FIT_BLOCK_NAMES]

    def sync_params(self):
        o = 0
        for k, n in zip(FIT_BLOCK_NAMES, self.sizes):
            self.pset.params[k].set_free(self.x[o:o + n])
            o += n
        self._written = [self.pset.params[k].value.copy() for k in FIT_BLOCK_NAMES]


def save_checkpoint(pset, path):
    np.savez(path, **{k.replace('/', '__'): q.value for k, q in pset.params.items()})
    return path if str(path).endswith('.npz') else str(path) + '.npz'


def load_checkpoint(pset, path, fitter=None, reset=True):
    """Restore the ParamSet from the .npz written by save_checkpoint.  fitter: a KronDeviceFit / zigp.optim.AdamGroups stepping this
    ParamSet -- it takes the loaded values as its state at once and, with reset=True (the default: a checkpoint holds parameters, not
    Adam moments -- tf.train.Saver restores them only if they were saved, onofftf/utils.py:61-73 saves all variables of the graph; ours
    does not keep them), restarts its moments and iteration count instead of carrying those of the parameters it replaced."""
    path = path if str(path).endswith('.npz') else str(path) + '.npz'
    d = np.load(path)
    for k, q in pset.params.items():
        q.value = np.asarray(d[k.replace('/', '__')], dtype=np.float64).reshape(q.value.shape)
    if fitter is not None:
        fitter.resync(reset=reset)
