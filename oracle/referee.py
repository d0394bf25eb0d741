"""Deterministic minimum-weight look-up referee ("static decoder").

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference's referee is a pre-trained Keras feed-forward homology-class
predictor whose weight blobs are absent from the checkout
(/root/reference/.MISSING_LARGE_BLOBS:1-4); it is consumed only through
``argmax(static_decoder.predict(true_syndrome))`` (Environments.py:144,150).
The README allows "any perfect-measurement decoding algorithm"
(README.md:278).  The build therefore defines its own referee, identically in
this oracle and in the HIP library:

  For one Pauli component (X: type-3 plaquettes, logical = parity over column 0,
  Function_Library.py:312-314;  Z: type-1 plaquettes, logical = parity over row
  0, Function_Library.py:315-317) let w_c(s) be the minimum number of
  single-qubit flips producing syndrome s AND logical class c.  The referee
  predicts class 1 iff w_1(s) < w_0(s)  (ties -> class 0).

The rule is independent of search order, so any breadth-first search over the
doubled space (syndrome, class) gives the same table.  Table bit order: bit i of
the index is the i-th plaquette of ``lattice.typed_order(d, type)``.
"""
import numpy as np
from . import lattice


def component_deltas(d, typ):
    """For each qubit: (syndrome-index delta, logical bit) of flipping that component."""
    typed = lattice.typed_order(d, typ)
    pos = {ab: i for i, ab in enumerate(typed)}
    n = len(typed)
    deltas = []
    for x in range(d):
        for y in range(d):
            s = 0
            for ab in ((x, y), (x, y + 1), (x + 1, y), (x + 1, y + 1)):
                if ab in pos:
                    s |= 1 << pos[ab]
            logical = (y == 0) if typ == 3 else (x == 0)
            deltas.append(s | (int(logical) << n))
    return n, deltas


def build_lut(d, typ):
    """uint8[2**n] table: predicted logical class of component `typ` (3 -> X part, 1 -> Z part)."""
    n, deltas = component_deltas(d, typ)
    size = 1 << (n + 1)
    dist = np.full(size, 255, dtype=np.uint8)
    dist[0] = 0
    frontier = np.zeros(1, dtype=np.int64)
    w = 0
    while frontier.size:
        w += 1
        for dl in deltas:
            cand = frontier ^ dl
            fresh = cand[dist[cand] == 255]
            dist[fresh] = w
        frontier = np.flatnonzero(dist == w)
    assert (dist != 255).all()
    half = 1 << n
    return (dist[half:] < dist[:half]).astype(np.uint8)


def build_ml_lut(d, typ, q):
    """uint8[2**n] maximum-likelihood table for independent component flips with probability q per qubit: the (syndrome, class)
    distribution is the XOR-convolution of the single-qubit ones, P'[s] = (1 - q) P[s] + q P[s ^ delta]; class 1 iff strictly more
    likely (the same arithmetic, in the same order, as csrc/env.hip ml_step_kernel)."""
    n, deltas = component_deltas(d, typ)
    size = 1 << (n + 1)
    p = np.zeros(size, dtype=np.float64)
    p[0] = 1.0
    idx = np.arange(size)
    q = np.float64(q)
    for dl in deltas:
        p = (np.float64(1.0) - q) * p + q * p[idx ^ dl]
    half = 1 << n
    return (p[half:] > p[:half]).astype(np.uint8)


class LutReferee:
    """Object with the ``predict`` signature the reference calls (Environments.py:144)."""

    def __init__(self, d, error_model, lut_x=None, lut_z=None):
        self.d = d
        self.error_model = error_model
        self.masks = lattice.Masks(d)
        self.lut_x = build_lut(d, 3) if lut_x is None else lut_x
        self.lut_z = build_lut(d, 1) if lut_z is None else lut_z
        self.n_classes = 2 if error_model == "X" else 4

    def classify_word(self, word):
        x = int(self.lut_x[self.masks.referee_index(word, 3)])
        if self.error_model == "X":
            return x
        z = int(self.lut_z[self.masks.referee_index(word, 1)])
        return x + 2 * z

    def predict(self, x, batch_size=1, verbose=0):
        x = np.asarray(x)
        out = np.zeros((x.shape[0], self.n_classes), dtype=np.float32)
        for i in range(x.shape[0]):
            grid = x[i].reshape(self.d + 1, self.d + 1)
            out[i, self.classify_word(self.masks.grid_to_word(grid))] = 1.0
        return out
