"""Static surface-code lattice tables, restated in closed form.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Paths below are relative to
/root/reference; ``FL`` = cluster_scripts/d5_dp/Function_Library.py,
``ENV`` = example_notebooks/Environments.py.

Geometry (FL:13-51): data qubit (x, y), 0 <= x, y < d, touches plaquettes
(x, y), (x, y+1), (x+1, y), (x+1, y+1) of the (d+1) x (d+1) plaquette grid.
Plaquette (a, b) has type 3 when a+b is odd, type 1 when even; it is absent
(type 0) on the four boundary patterns of FL:42-50.  A type-3 plaquette reports
the parity of the X-component of its qubits, a type-1 plaquette the parity of
the Z-component (FL:165-172: a qubit with error e toggles every touching
plaquette whose type is neither 0 nor e).
"""
import numpy as np


def plaquette_type(d, a, b):
    """FL:32-35 (type by parity) and FL:42-50 (absent boundary plaquettes)."""
    if (a == 0 and b % 2 == 0) or (a == d and b % 2 == 1) or (b == 0 and a % 2 == 1) or (b == d and a % 2 == 0):
        return 0
    return 3 if (a + b) % 2 == 1 else 1


def qubit_table(d):
    """FL:13-51 generateSurfaceCodeLattice: (d, d, 4, 3) int array of [a, b, type]."""
    if d % 2 != 1:
        raise Exception("for the surface code d must be odd!")
    q = np.zeros((d, d, 4, 3), dtype=np.int64)
    for x in range(d):
        for y in range(d):
            for k, (a, b) in enumerate(((x, y), (x, y + 1), (x + 1, y), (x + 1, y + 1))):
                q[x, y, k] = (a, b, plaquette_type(d, a, b))
    return q


def measurement_order(d):
    """Plaquettes in the order FL:189-221 (generate_faulty_syndrome) draws their
    measurement-error uniforms: bulk row-major, then row 0, row d, column 0, column d."""
    order = [(a, b) for a in range(1, d) for b in range(1, d)]
    half = (d + 1) // 2 - 1
    order += [(0, 2 * x + 1) for x in range(half)]
    order += [(d, 2 * x + 2) for x in range(half)]
    order += [(2 * x + 2, 0) for x in range(half)]
    order += [(2 * x + 1, d) for x in range(half)]
    return order


def typed_order(d, typ):
    """Live plaquettes of one type in row-major (a, b) order -- the bit order of
    the referee look-up index (bit i <-> i-th plaquette of this list)."""
    return [(a, b) for a in range(d + 1) for b in range(d + 1) if plaquette_type(d, a, b) == typ]


def plaquette_qubits(d, a, b):
    """Row-major indices of the (up to four) data qubits touching plaquette (a, b)."""
    out = []
    for x in (a - 1, a):
        for y in (b - 1, b):
            if 0 <= x < d and 0 <= y < d:
                out.append(x * d + y)
    return out


def qubit_stabilizers(d):
    """ENV:326-347: for each qubit (row-major) the list of live plaquettes it touches,
    in the (x,y),(x,y+1),(x+1,y),(x+1,y+1) order of FL:31-36."""
    out = []
    for x in range(d):
        for y in range(d):
            out.append([(a, b) for (a, b) in ((x, y), (x, y + 1), (x + 1, y), (x + 1, y + 1))
                        if plaquette_type(d, a, b) != 0])
    return out


def qubit_neighbours(d):
    """ENV:349-372: 8-neighbourhood (diagonals included, self excluded), in the
    product((0,-1,+1),(0,-1,+1)) order of ENV:360 minus its first element."""
    out = []
    for row in range(d):
        for col in range(d):
            cells = [(row + a, col + b) for a in (0, -1, 1) for b in (0, -1, 1)][1:]
            out.append([r * d + c for (r, c) in cells if 0 <= r < d and 0 <= c < d])
    return out


def identity_indicator(d):
    """ENV:374-385: ones everywhere except the qubit cells (2j+1, 2k+1)."""
    ind = np.ones((2 * d + 1, 2 * d + 1), dtype=np.int64)
    ind[1::2, 1::2] = 0
    return ind


def static_plane(d):
    """The syndrome-independent decoration written by padding_syndrome (ENV:284-298)."""
    n = 2 * d + 1
    out = np.zeros((n, n), dtype=np.int64)
    for x in range(n):
        for y in range(n):
            if (x == 0 or x == n - 1) and y % 2 == 1:
                out[x, y] = 1
            if (y == 0 or y == n - 1) and x % 2 == 1:
                out[x, y] = 1
            if x % 2 == 1 and y % 2 == 1 and (x + y) % 4 == 0:
                out[x, y] = 1
    return out


def num_actions(d, error_model, use_Y):
    """ENV:55-65 -> (num_actions, n_action_layers)."""
    if error_model == "X":
        return d * d + 1, 1
    if error_model in ("DP", "IIDXZ"):      # IIDXZ: the build's extension (the reference constructor rejects it) -- DP's action layers
        return (3 * d * d + 1, 3) if use_Y else (2 * d * d + 1, 2)
    raise ValueError("specified error model not currently supported!")


def layer_pauli(error_model, use_Y, layer):
    """FL:243-289 index_to_move: Pauli code applied by action layer `layer`."""
    if error_model == "X":
        return 1
    if use_Y:
        return layer + 1
    return 1 if layer == 0 else 3


class Masks:
    """Bit-mask view of the lattice used by the bit-plane restatement (oracle/env_oracle.py,
    oracle/env_oracle.c) and mirrored by the HIP kernel's constant tables.

    Qubit q = row*d + col is bit q of the x/z planes.  Stabilizers are numbered in
    measurement_order (bit s of a "syndrome word")."""

    def __init__(self, d):
        self.d = d
        self.order = measurement_order(d)
        self.n_stab = len(self.order)
        self.index = {ab: s for s, ab in enumerate(self.order)}
        self.stab_type = [plaquette_type(d, a, b) for (a, b) in self.order]
        assert all(t != 0 for t in self.stab_type) and self.n_stab == d * d - 1
        self.stab_qmask = [sum(1 << q for q in plaquette_qubits(d, a, b)) for (a, b) in self.order]
        # live plaquettes touched by each qubit, as a syndrome-word mask (ENV:262-271)
        self.qubit_smask = [sum(1 << self.index[ab] for ab in stabs) for stabs in qubit_stabilizers(d)]
        self.neigh_qmask = [sum(1 << n for n in ns) for ns in qubit_neighbours(d)]
        self.col0_mask = sum(1 << (x * d) for x in range(d))       # FL:312-314
        self.row0_mask = sum(1 << y for y in range(d))             # FL:315-317
        # referee index bit position of each stabilizer within its own type
        self.typed = {t: typed_order(d, t) for t in (1, 3)}
        self.ref_bit = [self.typed[self.stab_type[s]].index(ab) for s, ab in enumerate(self.order)]

    def syndrome_word(self, xmask, zmask):
        """FL:152-174 restated: bit s = parity of the matching component over the plaquette's qubits."""
        w = 0
        for s in range(self.n_stab):
            comp = xmask if self.stab_type[s] == 3 else zmask
            w |= (bin(comp & self.stab_qmask[s]).count("1") & 1) << s
        return w

    def word_to_grid(self, w):
        g = np.zeros((self.d + 1, self.d + 1), dtype=np.int64)
        for s, (a, b) in enumerate(self.order):
            g[a, b] = (w >> s) & 1
        return g

    def grid_to_word(self, g):
        return sum(int(g[a, b] != 0) << s for s, (a, b) in enumerate(self.order))

    def referee_index(self, w, typ):
        idx = 0
        for s in range(self.n_stab):
            if self.stab_type[s] == typ and (w >> s) & 1:
                idx |= 1 << self.ref_bit[s]
        return idx
