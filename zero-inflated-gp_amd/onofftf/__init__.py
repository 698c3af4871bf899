"""Look-alike of the reference's TensorFlow stack `onofftf` (+ the `onoff` fit function of scripts/onoff.py) on the
MI355X engine: same entry points and argument meaning, no TensorFlow / GPflow."""
from . import main  # noqa: F401
from .onoff import onoff  # noqa: F401
from .onoffpred import predict_onoff  # noqa: F401
