"""Minimal stand-ins for gpflow.param.{Param, DataHolder, Parameterized} as the reference uses them."""
import numpy as np

from zigp.optim import P
from zigp.transforms import Identity


class Param(P):
    """`.value` is the constrained value (what PlotOnOff1D reads, onoffgpf/PlotOnOff1D.py:16-26)."""

    def __init__(self, value, transform=None):
        v = np.atleast_1d(np.array(value, dtype=np.float64))
        super().__init__(v, transform or Identity())


class DataHolder:
    def __init__(self, array):
        self.value = np.asarray(array, dtype=np.float64)

    @property
    def shape(self):
        return self.value.shape


class Parameterized:
    """Assigning a number/array to an attribute that holds a Param sets its value
    (GPflow semantics used at zero-inflated-gpflow.ipynb:99-104,134)."""

    def __setattr__(self, key, val):
        cur = self.__dict__.get(key)
        if isinstance(cur, Param) and not isinstance(val, Param):
            cur.value = np.asarray(val, dtype=np.float64).reshape(cur.value.shape) if np.size(val) == cur.value.size \
                else np.atleast_1d(np.array(val, dtype=np.float64))
        else:
            object.__setattr__(self, key, val)
