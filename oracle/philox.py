"""Philox4x32-10 (Salmon et al., SC'11, "Parallel random numbers: as easy as 1, 2, 3").

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference never seeds its RNG and pulls scalars from numpy's global
MT19937 in a data-dependent order (Function_Library.py:99-100, 191).  Bit-exact
parity therefore needs an injected, site-indexed stream:

    W(seed, env_id, round, lane)[k] = Philox4x32-10(key=seed,
                                                   ctr=(round_lo, round_hi, env_id, lane | stream<<16))[k]

    stream 0 (environment noise), per syndrome-measurement round:
        k=0  uniform deciding whether qubit `lane` (row-major) suffers an error
        k=1  Pauli type of that error: 1 + ((W*3) >> 32)  in {1,2,3}
        k=2  uniform deciding whether stabilizer #`lane` (in the draw order of
             Function_Library.py:189-221) is mis-measured
    "u < p"  <=>  W < ceil(p * 2**32)   (W / 2**32 is exact in float64).
"""
import numpy as np

M0 = 0xD2511F53
M1 = 0xCD9E8D57
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = 0xFFFFFFFF

STREAM_ENV = 0      # environment noise (error + measurement draws)
STREAM_POLICY = 1   # epsilon-greedy / uniform-over-legal action draws
STREAM_REPLAY = 2   # replay minibatch index draws
STREAM_DROPOUT = 3  # dropout masks
STREAM_INIT = 4     # weight initialisation


def philox4x32(ctr, key, rounds=10):
    """Scalar Philox4x32.  ctr: 4 ints, key: 2 ints -> tuple of 4 uint32 (python ints)."""
    c0, c1, c2, c3 = (int(x) & MASK for x in ctr)
    k0, k1 = (int(x) & MASK for x in key)
    for _ in range(rounds):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> 32, p0 & MASK
        hi1, lo1 = p1 >> 32, p1 & MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & MASK, lo1, (hi0 ^ c3 ^ k1) & MASK, lo0
        k0 = (k0 + W0) & MASK
        k1 = (k1 + W1) & MASK
    return c0, c1, c2, c3


def philox4x32_np(c0, c1, c2, c3, key, rounds=10):
    """Vectorised Philox4x32 over broadcastable uint32 counter arrays -> 4 uint32 arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(*(np.asarray(x, dtype=np.uint64) & MASK for x in (c0, c1, c2, c3)))
    c0, c1, c2, c3 = c0.copy(), c1.copy(), c2.copy(), c3.copy()
    k0 = np.uint64(int(key[0]) & MASK)
    k1 = np.uint64(int(key[1]) & MASK)
    m = np.uint64(MASK)
    s32 = np.uint64(32)
    for _ in range(rounds):
        p0 = np.uint64(M0) * c0
        p1 = np.uint64(M1) * c2
        hi0, lo0 = p0 >> s32, p0 & m
        hi1, lo1 = p1 >> s32, p1 & m
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & m, lo1, (hi0 ^ c3 ^ k1) & m, lo0
        k0 = (k0 + np.uint64(W0)) & m
        k1 = (k1 + np.uint64(W1)) & m
    return tuple(x.astype(np.uint32) for x in (c0, c1, c2, c3))


def site_words(seed, env_id, rnd, lane, stream=STREAM_ENV):
    """The four words of site (env_id, rnd, lane) of `stream` (scalar)."""
    rnd = int(rnd)
    return philox4x32((rnd & MASK, (rnd >> 32) & MASK, env_id, (lane & 0xFFFF) | (stream << 16)), seed)


def threshold(p):
    """Integer T such that (W / 2**32 < p) <=> (W < T) for every uint32 W."""
    import math
    t = math.ceil(float(p) * 4294967296.0)
    return max(0, min(t, 1 << 32))


def threshold16(p):
    """Integer T such that (H / 2**16 < p) <=> (H < T) for every 16-bit draw H (dropout: csrc/common.h dq_rate_threshold16)."""
    import math
    t = math.ceil(float(p) * 65536.0)
    return max(0, min(t, 1 << 16))


def pauli_type(word):
    """Map a uint32 to {1,2,3} (X,Y,Z); stands in for np.random.randint(1,4) (Function_Library.py:100)."""
    return 1 + ((int(word) * 3) >> 32)


def bounded(word, n):
    """Map a uint32 to [0, n) by multiply-shift."""
    return (int(word) * int(n)) >> 32
