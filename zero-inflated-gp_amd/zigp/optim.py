"""Optimiser loops around the ELBO step (host side).

* `ParamSet` maps named constrained parameters <-> one flat free-state vector (the role of GPflow's
  Parameterized.get_free_state / set_state and of onofftf.main.Param, onofftf/main.py:137-184).
* `lbfgsb` is GPflow 0.4.0 Model.optimize's default (scipy.optimize.minimize(method='L-BFGS-B', jac=True)).
* `AdamGroups` reproduces scripts/onoff.py:325-350: one tf.train.AdamOptimizer per distinct learning
  rate (TF defaults beta1=0.9, beta2=0.999, eps=1e-8; update lr_t = lr*sqrt(1-b2^t)/(1-b1^t)).
"""
import numpy as np

from .transforms import Identity


class P:
    """One trainable (or fixed) parameter: constrained value + transform (+ learning rate for Adam groups)."""

    def __init__(self, value, transform=None, fixed=False, learning_rate=0.001, name=None):
        self.transform = transform or Identity()
        self.value = np.array(value, dtype=np.float64)
        self.fixed = fixed
        self.learning_rate = learning_rate
        self.name = name

    @property
    def shape(self):
        return self.value.shape

    def free(self):
        return np.asarray(self.transform.backward(self.value), dtype=np.float64).reshape(-1)

    def set_free(self, x):
        self.value = np.asarray(self.transform.forward(x), dtype=np.float64).reshape(self.value.shape)


class ParamSet:
    def __init__(self, params):
        """params: ordered dict name -> P"""
        self.params = params

    def names(self, trainable_only=True):
        return [k for k, p in self.params.items() if not (trainable_only and p.fixed)]

    def get_free(self):
        xs = [self.params[k].free() for k in self.names()]
        return np.concatenate(xs) if xs else np.zeros(0)

    def set_free(self, x):
        o = 0
        for k in self.names():
            p = self.params[k]
            n = p.value.size
            p.set_free(np.asarray(x[o:o + n]))
            o += n

    def values(self):
        return {k: p.value for k, p in self.params.items()}

    def free_grad(self, grads):
        """Chain constrained gradients (dict name -> array) to the flat free-state gradient."""
        gs = []
        for k in self.names():
            p = self.params[k]
            x = p.free()
            gs.append(np.asarray(p.transform.grad_free(x, np.asarray(grads[k], dtype=np.float64).reshape(-1))).reshape(-1))
        return np.concatenate(gs) if gs else np.zeros(0)


def lbfgsb(pset, value_and_grad, maxiter=1000, disp=False, callback=None, ftol=2.220446049250313e-09, gtol=1e-5):
    """Minimise -ELBO.  value_and_grad(values dict) -> (elbo, grads dict w.r.t. constrained values)."""
    from scipy.optimize import minimize

    def obj(x):
        pset.set_free(x)
        elbo, g = value_and_grad(pset.values())
        return -elbo, -pset.free_grad(g)

    res = minimize(obj, pset.get_free(), method='L-BFGS-B', jac=True, callback=callback,
                   options=dict(maxiter=maxiter, disp=disp, ftol=ftol, gtol=gtol))
    pset.set_free(res.x)
    return res


class AdamGroups:
    """One Adam per learning rate on the FREE state (TensorFlow's variables are the unconstrained values: scripts/onoff.py:88-123 builds
    the positive parameters as transforms of them, and the optimisers of :325-350 update the variables).  The free vectors are kept
    across steps (initialised from the constrained values once); every step writes the constrained values back into the ParamSet."""

    def __init__(self, pset, beta1=0.9, beta2=0.999, eps=1e-8):
        self.pset = pset
        self.b1, self.b2, self.eps = beta1, beta2, eps
        self.t = 0
        self.m = {k: np.zeros(pset.params[k].value.size) for k in pset.names()}
        self.v = {k: np.zeros(pset.params[k].value.size) for k in pset.names()}
        self.resync()

    def resync(self, reset=False):
        """Take the free vectors from the ParamSet again.  step() does this by itself for any parameter whose constrained value is no
        longer the one it wrote (load_checkpoint, an assignment to .value): the cached free vector would silently overwrite such a change.
        The moments and the iteration count are kept unless reset=True (parameters unrelated to the ones trained so far)."""
        if reset:
            self.t = 0
            for k in self.m:
                self.m[k][:] = 0.0
                self.v[k][:] = 0.0
        self.x = {k: self.pset.params[k].free().copy() for k in self.pset.names()}
        self._written = {k: self.pset.params[k].value.copy() for k in self.pset.names()}

    def step(self, grads):
        """One minimisation step of cost = -ELBO given d ELBO / d (constrained)."""
        self.t += 1
        for k in self.pset.names():
            p = self.pset.params[k]
            if k not in self.m:                                    # un-fixed since construction: it joins with fresh moments
                self.m[k], self.v[k] = np.zeros(p.value.size), np.zeros(p.value.size)
                self._written[k] = None
            if self._written[k] is None or not np.array_equal(p.value, self._written[k], equal_nan=True):   # changed behind our back: start from what the ParamSet holds now
                self.x[k] = p.free().copy()
            x = self.x[k]
            g = -np.asarray(p.transform.grad_free(x, np.asarray(grads[k], dtype=np.float64).reshape(-1))).reshape(-1)
            self.m[k] = self.b1 * self.m[k] + (1 - self.b1) * g
            self.v[k] = self.b2 * self.v[k] + (1 - self.b2) * g * g
            lr_t = p.learning_rate * np.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
            self.x[k] = x - lr_t * self.m[k] / (np.sqrt(self.v[k]) + self.eps)
            p.set_free(self.x[k])
            self._written[k] = p.value.copy()
