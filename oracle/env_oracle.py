"""Scalar CPU restatement of Surface_Code_Environment_Multi_Decoding_Cycles.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PINNED against
tests/golden/*.npz (made by tools/gen_golden.py from the reference itself).

``ENV`` = /root/reference/example_notebooks/Environments.py,
``FL``  = /root/reference/cluster_scripts/d5_dp/Function_Library.py.

State is held as bit-planes: Pauli codes I,X,Y,Z = 0,1,2,3 multiply as XOR
(FL:54-62 is the XOR table), so the hidden state is two qubit masks
``xmask`` (codes 1,2) and ``zmask`` (codes 2,3); syndromes are words over the
stabilizers in measurement order (oracle/lattice.py).
"""
import numpy as np
from . import lattice, philox


# ----------------------------------------------------------------------------------------
# stand-alone restatements of the Function_Library helpers (used by the golden-vector tests)
# ----------------------------------------------------------------------------------------

def codes_to_masks(error):
    e = np.asarray(error).astype(np.int64).reshape(-1)
    xm = sum(1 << q for q in range(e.size) if e[q] in (1, 2))
    zm = sum(1 << q for q in range(e.size) if e[q] in (2, 3))
    return xm, zm


def masks_to_codes(d, xm, zm):
    """Inverse of codes_to_masks: x -> 1, z -> 3, both -> 2."""
    out = np.zeros(d * d, dtype=np.int64)
    for q in range(d * d):
        x, z = (xm >> q) & 1, (zm >> q) & 1
        out[q] = (1 if x else 0) ^ (3 if z else 0)
    return out.reshape(d, d)


def syndrome_grid(error):
    """FL:152-174 generate_surface_code_syndrome_NoFT_efficient."""
    d = np.asarray(error).shape[0]
    m = lattice.Masks(d)
    return m.word_to_grid(m.syndrome_word(*codes_to_masks(error)))


def homology_label(error, err_model):
    """FL:296-326 generate_one_hot_labels_surface_code."""
    error = np.asarray(error)
    d = error.shape[0]
    m = lattice.Masks(d)
    xm, zm = codes_to_masks(error)
    X = bin(xm & m.col0_mask).count("1") & 1
    Z = bin(zm & m.row0_mask).count("1") & 1
    label = np.zeros(4 if err_model in ("IIDXZ", "DP") else 2, dtype=np.int64)
    label[X + 2 * Z] = 1
    return label


def index_to_move(d, move_index, error_model, use_Y=True):
    """FL:243-294: one-hot Pauli lattice of an action index (all-zero for the identity)."""
    out = np.zeros((d, d))
    _, layers = lattice.num_actions(d, error_model, use_Y)
    if move_index < layers * d * d:
        layer, q = divmod(move_index, d * d)
        out[q // d, q % d] = lattice.layer_pauli(error_model, use_Y, layer)
    return out


def padding_syndrome(d, syndrome):
    """ENV:273-299."""
    out = lattice.static_plane(d)
    out[0::2, 0::2] = np.asarray(syndrome)
    return out


def padding_actions(d, actions_in):
    """ENV:301-314."""
    out = np.zeros((2 * d + 1, 2 * d + 1), dtype=np.int64)
    for i, a in enumerate(actions_in):
        if a:
            out[2 * (i // d) + 1, 2 * (i % d) + 1] = 1
    return out


def iidxz_error_from_words(d, words, p_phys):
    """FL:134-160 generate_IIDXZ_error under injected 32-bit words: qubit q (row-major) draws words[2q] (X flip) then words[2q+1]
    (Z flip); X and Z together give Y (code 2)."""
    T = philox.threshold(p_phys)
    w = np.asarray(words, dtype=np.uint64).reshape(d, d, 2)
    x, z = w[..., 0] < T, w[..., 1] < T
    return (x * 1) ^ (z * 3)


def faulty_syndrome_from_words(d, true_grid, words, p_meas):
    """FL:176-223 generate_faulty_syndrome with the uniforms replaced by uint32 `words`
    (one per stabilizer, in measurement order)."""
    m = lattice.Masks(d)
    T = philox.threshold(p_meas)
    out = np.zeros((d + 1, d + 1), dtype=np.int64)
    for s, (a, b) in enumerate(m.order):
        flip = int(words[s]) < T
        out[a, b] = (1 - true_grid[a, b]) if flip else true_grid[a, b]
    return out


# ----------------------------------------------------------------------------------------
# the environment
# ----------------------------------------------------------------------------------------

class OracleEnv:
    """Bit-plane restatement of ENV:10-385 for ONE lattice, with the injected site RNG."""

    def __init__(self, d=5, p_phys=0.01, p_meas=0.01, error_model="DP", use_Y=True, volume_depth=3,
                 referee=None, seed=(0x5EED, 0xD0DEC0DE), env_id=0):
        self.d, self.p_phys, self.p_meas = d, p_phys, p_meas
        self.error_model, self.use_Y, self.volume_depth = error_model, use_Y, volume_depth
        self.num_actions, self.n_action_layers = lattice.num_actions(d, error_model, use_Y)   # ENV:55-65
        self.identity_index = self.num_actions - 1                                          # ENV:69
        self.m = lattice.Masks(d)
        self.referee = referee
        self.seed, self.env_id = seed, env_id
        self.round = 0                      # per-lattice measurement-round counter; never resets
        self.static = lattice.static_plane(d)
        C, n = volume_depth + self.n_action_layers, 2 * d + 1
        self.board_state = np.zeros((C, n, n), dtype=np.int64)
        self.xmask = self.zmask = 0
        self.true_word = 0
        self.summed_word = 0
        self.volume = [0] * volume_depth
        self.completed = 0                  # bit a <-> completed_actions[a]
        self.acted = 0                      # bit q <-> q in acted_on_qubits
        self.legal = 0                      # bit a <-> a in legal_actions
        self.done = False
        self.lifetime = 0

    # -- noise ------------------------------------------------------------------------
    def _draw_round(self):
        """One generate_error (FL:67-122) + one generate_faulty_syndrome (FL:176-223) worth of
        site words.  Returns (error xmask, error zmask, measurement flip word)."""
        d2 = self.d * self.d
        Tp, Tm = philox.threshold(self.p_phys), philox.threshold(self.p_meas)
        ex = ez = flips = 0
        # words of every lane of this round at once (same values as philox.site_words per lane)
        w0, w1, w2, _ = philox.philox4x32_np(self.round & philox.MASK, (self.round >> 32) & philox.MASK, self.env_id,
                                             np.arange(d2, dtype=np.uint32), self.seed)
        for lane in range(d2):
            if self.error_model == "IIDXZ":                 # FL:134-160: two uniforms per qubit, X flip first, then Z flip
                if int(w0[lane]) < Tp:
                    ex |= 1 << lane
                if int(w1[lane]) < Tp:
                    ez |= 1 << lane
            elif int(w0[lane]) < Tp:                        # FL:99 / FL:119
                t = 1 if self.error_model == "X" else philox.pauli_type(w1[lane])   # FL:100
                if t in (1, 2):
                    ex |= 1 << lane
                if t in (2, 3):
                    ez |= 1 << lane
            if lane < self.m.n_stab and int(w2[lane]) < Tm:  # FL:191,199,205,212,218
                flips |= 1 << lane
        self.round += 1
        return ex, ez, flips

    def _new_volume(self):
        """ENV:157-172 == ENV:216-231: draw volumes until one is non-trivial."""
        while True:
            self.summed_word = 0
            for j in range(self.volume_depth):
                ex, ez, flips = self._draw_round()
                self.xmask ^= ex                            # ENV:164 / FL:226-241 (XOR)
                self.zmask ^= ez
                self.true_word = self.m.syndrome_word(self.xmask, self.zmask)   # ENV:165
                self.volume[j] = self.true_word ^ flips     # ENV:166
                self.summed_word |= self.volume[j]          # ENV:168 (sum != 0 <=> OR != 0)
                self.lifetime += 1                          # ENV:169
            if self.summed_word:                            # ENV:171
                return

    # -- bookkeeping ---------------------------------------------------------------------
    def _reset_legal_moves(self):
        """ENV:238-258."""
        d2 = self.d * self.d
        self.completed = 0
        self.acted = 0
        self.legal = 1 << self.identity_index
        for q in range(d2):
            if self.m.qubit_smask[q] & self.summed_word:    # ENV:262-271
                for j in range(self.n_action_layers):
                    self.legal |= 1 << (q + j * d2)

    def _write_syndrome_planes(self):
        for j in range(self.volume_depth):                  # ENV:174-175
            self.board_state[j] = padding_syndrome(self.d, self.m.word_to_grid(self.volume[j]))

    def _write_action_planes(self):
        d2 = self.d * self.d
        for k in range(self.n_action_layers):               # ENV:200-201
            bits = [(self.completed >> (k * d2 + i)) & 1 for i in range(d2)]
            self.board_state[self.volume_depth + k] = padding_actions(self.d, bits)

    # -- gym protocol ----------------------------------------------------------------------
    def reset(self):
        """ENV:99-115 + ENV:206-235."""
        self.done = False
        self.lifetime = 0
        self.xmask = self.zmask = 0
        self.true_word = 0
        self.board_state[:] = 0
        self._new_volume()
        self._write_syndrome_planes()
        self._reset_legal_moves()
        return self.board_state

    def true_class(self):
        X = bin(self.xmask & self.m.col0_mask).count("1") & 1
        Z = bin(self.zmask & self.m.row0_mask).count("1") & 1
        return X + 2 * Z

    def step(self, action):
        """ENV:118-204."""
        action = int(action)
        d2 = self.d * self.d
        done_identity = action == self.identity_index or (self.completed >> action) & 1   # ENV:131
        if action < self.n_action_layers * d2:              # ENV:135-136
            layer, q = divmod(action, d2)
            pauli = lattice.layer_pauli(self.error_model, self.use_Y, layer)
            if pauli in (1, 2):
                self.xmask ^= 1 << q
            if pauli in (2, 3):
                self.zmask ^= 1 << q
        self.true_word = self.m.syndrome_word(self.xmask, self.zmask)     # ENV:139
        correct = self.true_class()                         # ENV:143
        decoded = self.referee.classify_word(self.true_word)    # ENV:144
        reward = 0.0
        if correct == 0 and self.true_word == 0:            # ENV:148-149
            reward = 1.0
        elif decoded != correct:                            # ENV:150-151
            self.done = True
        if done_identity:                                   # ENV:155-182
            self._new_volume()
            self._write_syndrome_planes()
            self._reset_legal_moves()
            self.board_state[self.volume_depth:] = 0
        else:                                               # ENV:185-201
            self.completed |= 1 << action
            q = action % d2
            if not (self.acted >> q) & 1:
                self.acted |= 1 << q
                for j in range(self.n_action_layers):
                    self.legal |= self.m.neigh_qmask[q] << (j * d2)
            self._write_action_planes()
        return self.board_state, reward, self.done, {}

    # -- views in the reference's own data types (for comparisons) ---------------------------
    @property
    def hidden_state(self):
        return masks_to_codes(self.d, self.xmask, self.zmask)

    @property
    def legal_actions(self):
        return {a for a in range(self.num_actions) if (self.legal >> a) & 1}

    @property
    def current_true_syndrome(self):
        return self.m.word_to_grid(self.true_word)
