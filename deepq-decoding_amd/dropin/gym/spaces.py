"""gym.spaces.Box / Discrete with the attributes the reference reads (shape, n, low, high, dtype)."""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low, self.high, self.dtype = low, high, dtype
        self.shape = tuple(shape) if shape is not None else tuple(np.shape(low))


class Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
