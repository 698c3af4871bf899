"""onofftf/utils.py look-alike: printtime (:4-8), kernse_np (:34-58, evaluated by libzigp's zigp_rbf_K), modelmanager (:61-73,
npz checkpoints of a ParamSet instead of tf.train.Saver)."""
import time

import numpy as np


def printtime(start):
    h, rem = divmod(time.time() - start, 3600)
    m, s = divmod(rem, 60)
    return '{:0>2}:{:0>2}:{:05.2f}'.format(int(h), int(m), s)


class kernse_np:
    def __init__(self, lengthscales, variance, engine=None):
        self.lengthscales, self.variance, self._engine = lengthscales, variance, engine

    def _eng(self):
        from onoffgpf.kernels import _get_engine
        return self._engine or _get_engine()

    def K(self, X, X2=None):
        D = np.shape(X)[1]
        ell = np.broadcast_to(np.asarray(self.lengthscales, dtype=np.float64).reshape(-1), (D,)) if np.size(self.lengthscales) in (1, D) \
            else np.asarray(self.lengthscales, dtype=np.float64)
        return self._eng().rbf_K(X, X2, np.ascontiguousarray(ell), float(np.squeeze(self.variance)))

    def Ksymm(self, X):
        return self.K(X)

    def Kdiag(self, X):
        return np.full(np.shape(X)[0], float(np.squeeze(self.variance)))


class modelmanager:
    """modelmanager(saver, sess, path): `saver` is the ParamSet (or anything with .params of Params); `sess` is ignored."""

    def __init__(self, saver, sess, path):
        self.saver, self.sess, self.path = saver, sess, path

    def save(self):
        from .model import save_checkpoint
        save_checkpoint(self.saver, self.path)
        print('model saved in : ' + str(self.path))

    def load(self):
        from .model import load_checkpoint
        load_checkpoint(self.saver, self.path)
        print('model loaded from : ' + str(self.path))
