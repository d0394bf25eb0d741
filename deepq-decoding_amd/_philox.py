"""Host-side Philox4x32-10 (numpy), used only for one-off initialisation (weight init) so that results are
reproducible across GPU counts.  The hot-path generators are in csrc/common.h."""
import numpy as np

_M0, _M1, _W0, _W1, _MASK = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF


def philox4x32(c0, c1, c2, c3, key):
    c0, c1, c2, c3 = np.broadcast_arrays(*(np.asarray(x, dtype=np.uint64) & _MASK for x in (c0, c1, c2, c3)))
    c0, c1, c2, c3 = c0.copy(), c1.copy(), c2.copy(), c3.copy()
    k0, k1 = np.uint64(int(key[0]) & _MASK), np.uint64(int(key[1]) & _MASK)
    m, s = np.uint64(_MASK), np.uint64(32)
    for _ in range(10):
        p0, p1 = np.uint64(_M0) * c0, np.uint64(_M1) * c2
        c0, c1, c2, c3 = ((p1 >> s) ^ c1 ^ k0) & m, p1 & m, ((p0 >> s) ^ c3 ^ k1) & m, p0 & m
        k0, k1 = (k0 + np.uint64(_W0)) & m, (k1 + np.uint64(_W1)) & m
    return c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32)
