"""gpflow.kernels.RBF look-alike (zero-inflated-gpflow.ipynb:98-104; arithmetic twin KernSE onofftf/main.py:33-63)."""
import numpy as np

from zigp.transforms import positive
from .param import Param, Parameterized

_engine = None


def _get_engine():
    global _engine
    if _engine is None:
        import zigp
        _engine = zigp.reference_engine(0)
    return _engine


class RBF(Parameterized):
    def __init__(self, input_dim, variance=1.0, lengthscales=None, active_dims=None, ARD=False):
        self.input_dim = int(input_dim)
        self.ARD = bool(ARD)
        self.variance = Param(variance, positive)
        if lengthscales is None:
            lengthscales = np.ones(self.input_dim) if ARD else 1.0
        self.lengthscales = Param(lengthscales, positive)

    def ell_vector(self):
        l = self.lengthscales.value.reshape(-1)
        return np.full(self.input_dim, l[0]) if l.size == 1 else l

    def compute_K(self, X, X2):
        return _get_engine().rbf_K(X, X2, self.ell_vector(), float(self.variance.value.reshape(-1)[0]))

    def compute_K_symm(self, X):
        return self.compute_K(X, None)

    def compute_Kdiag(self, X):
        return np.full(np.shape(X)[0], float(self.variance.value.reshape(-1)[0]))
