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
    """Minibatch iterator with the batch sequence of onofftf/main.py:66-133: numpy.random.seed(121) (random_seed.get_seed(121) -> op seed
    121, :70-73), one shuffle before the first batch, one at every epoch end, and a wrap-around batch made of the old epoch's tail plus the
    new epoch's head.  Kept as an ORDER (an index permutation of the caller's rows; every reshuffle composes onto it) instead of
    re-gathered copies of the data: `next_indices` is the iterator, `next_batch` gathers, and `next_span` tells a host that keeps the
    permuted epoch resident on the GPU (zigp_kron_elbo_rows) which row range of it the next batch is."""

    def __init__(self, xtrain, ytrain, dtype=None, seed=121):
        numpy.random.seed(seed)                      # the global generator, as the reference (its later draws interleave with the fit's)
        self._x, self._y = xtrain, ytrain
        self._n = xtrain.shape[0]
        self._order = numpy.arange(self._n)          # row of the caller's arrays at each position of the current epoch
        self._pos = 0
        self._epochs_completed = 0
        self._shuffles = 0                           # generation of `_order` (what a resident copy of the epoch is tagged with)
        self._cache = None

    def _reshuffle(self):
        perm = numpy.arange(self._n)
        numpy.random.shuffle(perm)
        self._order = self._order[perm]
        self._shuffles += 1
        self._cache = None

    # -- the reference's read-only views: the data in the current epoch's order
    @property
    def xtrain(self):
        return self._epoch_arrays()[0]

    @property
    def ytrain(self):
        return self._epoch_arrays()[1]

    def _epoch_arrays(self):
        if self._cache is None:
            self._cache = (self._x[self._order], self._y[self._order])
        return self._cache

    @property
    def num_examples(self):
        return self._n

    @property
    def epochs_completed(self):
        return self._epochs_completed

    @property
    def generation(self):
        return self._shuffles

    def _advance(self, batch_size, shuffle):
        """-> (generation, lo, hi, tail): positions [lo, hi) of the current order; tail = row indices of the previous epoch's rest that
        precede them in a wrap-around batch (None otherwise)"""
        if self._epochs_completed == 0 and self._pos == 0 and shuffle and self._shuffles == 0:
            self._reshuffle()                                        # :102-107
        lo = self._pos
        if lo + batch_size > self._n:                                # :110-129
            self._epochs_completed += 1
            tail = self._order[lo:self._n].copy()
            if shuffle:
                self._reshuffle()
            self._pos = batch_size - (self._n - lo)
            return self._shuffles, 0, self._pos, tail
        self._pos = lo + batch_size                                  # :130-133
        return self._shuffles, lo, self._pos, None

    def next_indices(self, batch_size, shuffle=True):
        """rows of the caller's (xtrain, ytrain) that make up the next batch"""
        _, lo, hi, tail = self._advance(batch_size, shuffle)
        idx = self._order[lo:hi]
        return idx if tail is None else numpy.concatenate((tail, idx))

    def next_batch(self, batch_size, shuffle=True):
        idx = self.next_indices(batch_size, shuffle)
        return self._x[idx], self._y[idx]

    def next_span(self, batch_size, shuffle=True):
        """-> (generation, lo, hi, None) when the next batch is rows [lo, hi) of the current epoch's arrays (`xtrain`, `ytrain`: upload
        them once per generation), or (generation, None, None, (xb, yb)) for the one wrap-around batch per epoch"""
        gen, lo, hi, tail = self._advance(batch_size, shuffle)
        if tail is None:
            return gen, lo, hi, None
        idx = numpy.concatenate((tail, self._order[lo:hi]))
        return gen, None, None, (self._x[idx], self._y[idx])
