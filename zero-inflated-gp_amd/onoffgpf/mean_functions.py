"""gpflow.mean_functions look-alikes for `OnOffSVGP(..., mean_function=...)` (onoffgpf/OnOffSVGP.py:26,29,134):
Zero (the reference default), Constant(c) and Linear(A, b) with one output column (num_latent is 1, OnOffSVGP.py:45).
All three are the linear form m(x) = b + a . x, which libzigp evaluates inside the point-wise kernel
(zigp_set_mean_function, include/zigp.h); `__call__` gives the same values on the host for plotting."""
import numpy as np

from .param import Param, Parameterized


class MeanFunction(Parameterized):
    def linear_form(self, D):
        """(a (D,) or None, b float or None) for the engine's params dict, and the Params that hold them."""
        raise NotImplementedError

    def __call__(self, X):
        a, b = self.linear_form(np.shape(X)[1])
        m = np.zeros((np.shape(X)[0], 1))
        if a is not None:
            m = m + np.asarray(X, dtype=np.float64) @ a.reshape(-1, 1)
        if b is not None:
            m = m + b
        return m


class Zero(MeanFunction):
    def linear_form(self, D):
        return None, None

    def trainables(self):
        return {}


class Constant(MeanFunction):
    def __init__(self, c=None):
        self.c = Param(np.zeros(1) if c is None else c)

    def linear_form(self, D):
        return None, float(self.c.value.reshape(-1)[0])

    def trainables(self):
        return {'mean_b': self.c}


class Linear(MeanFunction):
    def __init__(self, A=None, b=None):
        A = np.ones((1, 1)) if A is None else np.atleast_2d(np.array(A, dtype=np.float64))
        if A.shape[1] != 1:
            raise ValueError('Linear mean function: A must be (D, 1) -- the model has one latent output')
        self.A = Param(A)
        self.b = Param(np.zeros(1) if b is None else b)

    def linear_form(self, D):
        a = self.A.value.reshape(-1)
        if a.size != D:
            raise ValueError('Linear mean function: A has %d rows, the inputs have %d columns' % (a.size, D))
        return a, float(self.b.value.reshape(-1)[0])

    def trainables(self):
        return {'mean_a': self.A, 'mean_b': self.b}
