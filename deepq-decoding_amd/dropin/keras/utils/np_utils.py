import numpy as np


def to_categorical(y, num_classes=None):
    y = np.asarray(y, dtype=int).ravel()
    n = int(num_classes) if num_classes else int(y.max()) + 1
    out = np.zeros((y.size, n), dtype=np.float32)
    out[np.arange(y.size), y] = 1.0
    return out
