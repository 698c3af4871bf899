"""Drop-in look-alike of the reference package `onoffgpf` (onoffgpf/__init__.py:1-4) on the MI355X engine.
No GPflow, no TensorFlow: the graph is replaced by libzigp.so (include/zigp.h)."""
from .OnOffSVGP import OnOffSVGP
from .OnOffLikelihood import OnOffLikelihood
from . import kernels
from . import mean_functions

__all__ = ['OnOffSVGP', 'OnOffLikelihood', 'kernels', 'mean_functions']
