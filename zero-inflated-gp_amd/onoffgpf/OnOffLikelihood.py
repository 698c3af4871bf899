"""OnOffLikelihood look-alike (onoffgpf/OnOffLikelihood.py:12-32): holds the noise variance (positive, 0.01)."""
import numpy as np

from zigp.transforms import positive
from .param import Param, Parameterized


class OnOffLikelihood(Parameterized):
    def __init__(self):
        self.variance = Param(0.01, positive)      # OnOffLikelihood.py:26

    def variational_expectations(self, Fmu, Fvar, Fmuvar, Y):
        """Closed form of OnOffLikelihood.py:30-32 on host arrays (for inspection; the engine fuses it)."""
        v = float(self.variance.value.reshape(-1)[0])
        return -0.5 * np.log(2 * np.pi) - 0.5 * np.log(v) - 0.5 * (np.square(Y - Fmu) + Fvar + Fmuvar) / v
