"""onofftf.main look-alike: Param, KernSE, DataSet (onofftf/main.py:33-184).  The graph-building helpers
(GaussKL, GPConditional, GaussKLkron, tf_kron) are replaced by libzigp.so entry points (include/zigp.h)."""
import numpy

from zigp.optim import P
from zigp.transforms import Identity

jitter_level = 1e-4   # onofftf/main.py:11 (only used by the reference's dead code paths)


class Param(P):
    """onofftf/main.py:137-184: value + transform (+ fixed flag, name, learning rate); get_tfv() -> constrained value."""

    def __init__(self, value, transform=None, fixed=False, name=None, learning_rate=None, summ=False):
        super().__init__(numpy.atleast_1d(numpy.array(value, dtype=numpy.float64)), transform or Identity(), fixed=fixed,
                         learning_rate=0.001 if learning_rate is None else learning_rate,   # Variable.learning_rate default, main.py:25-31
                         name=name or 'param')

    def get_tfv(self):
        return self.value

    def get_optv(self):
        return self.free()


class KernSE:
    """onofftf/main.py:33-63 (squared-exponential kernel of one Kronecker factor)."""

    def __init__(self, lengthscales, variance):
        self.lengthscales, self.variance = lengthscales, variance

    def _vals(self):
        l = self.lengthscales.get_tfv() if hasattr(self.lengthscales, 'get_tfv') else numpy.asarray(self.lengthscales)
        v = self.variance.get_tfv() if hasattr(self.variance, 'get_tfv') else numpy.asarray(self.variance)
        return numpy.asarray(l, dtype=numpy.float64).reshape(-1), float(numpy.asarray(v).reshape(-1)[0])

    def K(self, X, X2=None, engine=None):
        from onoffgpf.kernels import _get_engine
        l, v = self._vals()
        return (engine or _get_engine()).rbf_K(X, X2, l, v)

    def Ksymm(self, X):
        return self.K(X)

    def Kdiag(self, X):
        return numpy.full(numpy.shape(X)[0], self._vals()[1])


class DataSet(object):
    """Minibatch iterator, onofftf/main.py:66-133: numpy.random.seed(121) (random_seed.get_seed(121) -> op seed 121, :70-73),
    shuffle at the first call, reshuffle at every epoch end, wrap-around batches concatenate rest + new part."""

    def __init__(self, xtrain, ytrain, dtype=None, seed=121):
        numpy.random.seed(seed)
        self._num_examples = xtrain.shape[0]
        self._xtrain, self._ytrain = xtrain, ytrain
        self._epochs_completed = 0
        self._index_in_epoch = 0

    @property
    def xtrain(self):
        return self._xtrain

    @property
    def ytrain(self):
        return self._ytrain

    @property
    def num_examples(self):
        return self._num_examples

    @property
    def epochs_completed(self):
        return self._epochs_completed

    def next_batch(self, batch_size, shuffle=True):
        start = self._index_in_epoch
        if self._epochs_completed == 0 and start == 0 and shuffle:          # :102-107
            perm0 = numpy.arange(self._num_examples)
            numpy.random.shuffle(perm0)
            self._xtrain, self._ytrain = self.xtrain[perm0], self.ytrain[perm0]
        if start + batch_size > self._num_examples:                          # :110-129
            self._epochs_completed += 1
            rest = self._num_examples - start
            x_rest, y_rest = self._xtrain[start:self._num_examples], self._ytrain[start:self._num_examples]
            if shuffle:
                perm = numpy.arange(self._num_examples)
                numpy.random.shuffle(perm)
                self._xtrain, self._ytrain = self.xtrain[perm], self.ytrain[perm]
            start = 0
            self._index_in_epoch = batch_size - rest
            end = self._index_in_epoch
            return (numpy.concatenate((x_rest, self._xtrain[start:end]), axis=0),
                    numpy.concatenate((y_rest, self._ytrain[start:end]), axis=0))
        self._index_in_epoch += batch_size                                   # :130-133
        end = self._index_in_epoch
        return self._xtrain[start:end], self._ytrain[start:end]
