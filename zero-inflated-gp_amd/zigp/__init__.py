"""zigp -- MI355X-native zero-inflated ("OnOff") sparse variational GP engine (host side).

Python here is plumbing over the C-ABI in include/zigp.h (libzigp.so, hand-written HIP for gfx950).
There is no CPU fallback: importing works anywhere, using an engine needs the built library and a GPU.
"""
from ._lib import ZigpError, NotPositiveDefiniteError  # noqa: F401
from .engine import DenseEngine, PARAM_KEYS, reference_engine  # noqa: F401
