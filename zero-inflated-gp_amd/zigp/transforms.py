"""Parameter transforms of the reference's Param objects (host-side chain rule).

GPflow 0.4.0 `transforms.positive` = `Log1pe` (un-vendored; used at onoffgpf/OnOffSVGP.py:61,63,
onoffgpf/OnOffLikelihood.py:26, scripts/onoff.py:88-123):  y = log(1 + exp(x)) + 1e-6,
x = ys + log(-expm1(-ys)) with ys = max(y - 1e-6, eps).  [GPflow-recall; SURVEY.md a9]
"""
import numpy as np


class Identity:
    def forward(self, x):
        return x

    def backward(self, y):
        return y

    def grad_free(self, x, dy):
        """dL/dx given dL/dy (y = forward(x))."""
        return dy

    def __repr__(self):
        return 'Identity'


class Log1pe:
    def __init__(self, lower=1e-6):
        self._lower = lower

    def forward(self, x):
        x = np.asarray(x, dtype=np.float64)
        return np.logaddexp(0.0, x) + self._lower          # overflow-safe softplus

    def backward(self, y):
        ys = np.maximum(np.asarray(y, dtype=np.float64) - self._lower, np.finfo(np.float64).eps)
        return ys + np.log(-np.expm1(-ys))

    def grad_free(self, x, dy):
        x = np.asarray(x, dtype=np.float64)
        return np.asarray(dy) * (0.5 * (1.0 + np.tanh(0.5 * x)))   # sigmoid(x)

    def __repr__(self):
        return '+ve'


positive = Log1pe()
