"""`predict_scgp` -- onofftf/svcppred.py:15-224 (Bernoulli / probit Kronecker classifier): returns {'pfmean','pfvar'}
(class-1 probability probit(fmean / sqrt(1 + fvar)) and its Bernoulli variance) for train (and test)."""
import numpy as np

from .heads import restore_and_predict

jitter_level = 1e-6   # onofftf/svcppred.py:13


def predict_scgp(Xtrain, Xtest, checkpointPath, num_inducing_f=np.array([10, 100]), include_f_mu=False, device=0, engine=None):
    return restore_and_predict('bernoulli', Xtrain, Xtest, checkpointPath, num_inducing_f, include_f_mu,
                               (('pfmean', 2), ('pfvar', 3)), device=device, engine=engine)
