"""scripts/onoff.py: `onoff(Xtrain, Ytrain, Xtest, Ytest, dir)` -- see onofftf/onoff.py."""
from onofftf.onoff import onoff, jitter_level  # noqa: F401
