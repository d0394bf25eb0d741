"""Stand-in for `from keras.optimizers import Adam` (the agent only needs the hyper-parameters)."""
from _bootstrap import package as _package

Adam = _package("agent").Adam
