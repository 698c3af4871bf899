"""`predict_svgp` -- onofftf/svgppred.py:15-203 (Gaussian-likelihood Kronecker SVGP): returns {'fmean','fvar'} for the
training inputs (and the test inputs when given)."""
import numpy as np

from .heads import restore_and_predict

jitter_level = 1e-6   # onofftf/svgppred.py:13


def predict_svgp(Xtrain, Xtest, checkpointPath, num_inducing_f=np.array([10, 100]), include_fmu=False, device=0, engine=None):
    # include_fmu only defines an unused initial value in the reference (svgppred.py:26-27)
    return restore_and_predict('gaussian', Xtrain, Xtest, checkpointPath, num_inducing_f, False,
                               (('fmean', 0), ('fvar', 1)), device=device, engine=engine)       # :180-186
