"""`predict_onoff` -- onofftf/onoffpred.py:15-286: rebuild the parameter set, restore the checkpoint, predict with
jitter 1e-6 (:13) and gmean shifted by -1 (:141); returns {'gfmean','fmean','pgmean'} for train (and test)."""
import os

import numpy as np

import zigp
from .model import init_params, engine_params, load_checkpoint

jitter_level = 1e-6   # onofftf/onoffpred.py:13


def predict_onoff(Xtrain, Xtest, checkpointPath, num_inducing_f=np.array([10, 100]), num_inducing_g=np.array([10, 100]),
                  include_fmu=False, device=0, engine=None):
    # include_fmu has no effect in the reference either: its f_mu Param is commented out (onoffpred.py:27-28,89)
    pset = init_params(Xtrain, num_inducing_f, num_inducing_g, init_noisevar=0.001)
    ck = checkpointPath
    if os.path.isdir(ck):
        ck = os.path.join(ck, 'model')
    load_checkpoint(pset, ck)
    eng = engine or zigp.reference_engine(device)      # tf.cholesky's acceptance rule (pivot > 0)
    p = engine_params(pset)

    def run(X):
        o = eng.kron_predict(p, X, jitter=jitter_level, g_offset=-1.0)
        return {'gfmean': o[0].reshape(-1, 1), 'fmean': o[3].reshape(-1, 1), 'pgmean': o[7].reshape(-1, 1)}   # :273-280

    pred_train = run(Xtrain)
    if Xtest is not None:
        return pred_train, run(Xtest)
    return pred_train
