"""Feed-forward referee decoders with the reference's protocol.

The reference hands the environment a Keras model as ``static_decoder`` -- "a fast feed-forward NN homology class predictor"
(/root/reference/example_notebooks/Environments.py:53, loaded with ``load_model`` at
cluster_scripts/d5_dp/0.001/Single_Point_Training_Script.py:54-57) -- and calls
``static_decoder.predict(syndrome.reshape(1, (d+1)**2), batch_size=1, verbose=0)`` once per step, using only the argmax
(Environments.py:144,150).  The shipped blobs are missing (.MISSING_LARGE_BLOBS), so this module provides the same thing
from a weight file: a stack of Dense layers (ReLU between, softmax on top) read from a Keras HDF5 / .npz weight file.

On the device a referee is either tabulated -- ``VectorEnv.set_referee_predict`` calls ``predict`` once over all 2**n_stab
syndromes and installs the kernel's joint look-up table (d <= 5: one table read per step whatever the network's size) -- or, where no
table fits (d = 7: 48 stabilizers), EVALUATED there: ``VectorEnv.set_referee_mlp`` hands the Dense stack to
``dq_env_set_referee_mlp`` (include/deepq_hip.h; csrc/env.hip referee_mlp_kernel, one wavefront per lattice, before every step).
``FeedForwardReferee.predict_exact`` restates the device's fixed float32 arithmetic, so the two give the same classes bit for bit.
"""
import ctypes

import numpy as np

from .weights_io import load_weights_file


class FeedForwardReferee:
    """Dense(n_1, relu) ... Dense(n_classes, softmax) on the flattened (d+1) x (d+1) syndrome."""

    def __init__(self, weights):
        """weights: [kernel_1, bias_1, kernel_2, bias_2, ...] in Keras order and shapes (kernels (in, out))."""
        assert len(weights) >= 2 and len(weights) % 2 == 0
        self.layers = [(np.asarray(weights[i], np.float32), np.asarray(weights[i + 1], np.float32)) for i in range(0, len(weights), 2)]
        for (k, b), (k2, _) in zip(self.layers, self.layers[1:]):
            assert k.ndim == 2 and b.shape == (k.shape[1],) and k2.shape[0] == k.shape[1], "not a Dense stack"
        self.n_inputs, self.n_classes = self.layers[0][0].shape[0], self.layers[-1][0].shape[1]

    @classmethod
    def from_file(cls, path):
        """A Keras 2.x ``save_weights`` HDF5 file (or the package's .npz container) holding only Dense layers."""
        return cls(load_weights_file(path))

    def predict(self, x, batch_size=None, verbose=0):
        h = np.asarray(x, dtype=np.float32).reshape(len(x), -1)
        assert h.shape[1] == self.n_inputs, (h.shape, self.n_inputs)
        for i, (k, b) in enumerate(self.layers):
            h = h @ k + b
            if i + 1 < len(self.layers):
                np.maximum(h, 0.0, out=h)
        h -= h.max(axis=1, keepdims=True)
        e = np.exp(h)
        return e / e.sum(axis=1, keepdims=True)

    @property
    def dims(self):
        return [self.n_inputs] + [k.shape[1] for k, _ in self.layers]

    def flat_weights(self):
        """The stack as dq_env_set_referee_mlp takes it: per layer the kernel (in, out) row-major, then the bias; float32."""
        return np.concatenate([np.concatenate([k.reshape(-1), b]) for k, b in self.layers]).astype(np.float32)

    def logits_exact(self, x):
        """The device kernel's arithmetic (csrc/env.hip referee_mlp_kernel), restated: float32 throughout, accumulator = bias, then the
        inputs in increasing index order with one rounded multiply and one rounded add each, ReLU between layers.  (An exactly-zero
        input contributes +0.0 * w = 0: skipping it, as the kernel does for the binary first layer, leaves the same bits.)"""
        h = np.asarray(x, dtype=np.float32).reshape(len(x), -1)
        assert h.shape[1] == self.n_inputs, (h.shape, self.n_inputs)
        for i, (k, b) in enumerate(self.layers):
            acc = np.broadcast_to(b, (len(h), len(b))).astype(np.float32).copy()
            for j in range(k.shape[0]):
                col = h[:, j]
                if not col.any():
                    continue
                acc = acc + (col[:, None] * k[j][None, :]).astype(np.float32)      # float32 * float32 and float32 + float32: one rounding each
            h = np.maximum(acc, np.float32(0.0)) if i + 1 < len(self.layers) else acc
        return h

    def predict_exact(self, x, batch_size=None, verbose=0):
        """One-hot rows of the first maximum of logits_exact: what the device referee decides."""
        z = self.logits_exact(x)
        out = np.zeros_like(z)
        out[np.arange(len(z)), np.argmax(z, axis=1)] = 1.0
        return out


class MatchingReferee:
    """The exact minimum-weight referee for lattices too large for look-up tables (d >= 9; any odd 3 <= d <= 15), on the GPU:
    include/deepq_hip.h ``dq_match_*`` (csrc/match.hip).  Same definition as the environment's built-in look-up referee -- class 1 iff the
    lightest error with that syndrome and class 1 is strictly lighter than with class 0 -- and the same ``predict`` protocol the reference
    calls (Environments.py:144): rows of flattened (d+1) x (d+1) syndromes in, one-hot class rows out."""

    def __init__(self, d, error_model="DP"):
        from . import _lib
        import torch
        self._lib, self._torch = _lib, torch
        self.d, self.error_model = int(d), error_model
        self.n_classes = 2 if error_model == "X" else 4
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().dq_match_create(self.d, ctypes.byref(h)))
        self._h = h
        n, k, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.lib().dq_match_info(h, ctypes.byref(n), ctypes.byref(k), ctypes.byref(w)))
        self.nodes, self.max_defects, self.distance = n.value, k.value, w.value
        # plaquette (a, b) -> (component, bit) in row-major order per type (FL:32-35, 42-50: type 3 when a + b is odd)
        self._where = {}
        rank = [0, 0]
        dd = self.d
        for a in range(dd + 1):
            for b in range(dd + 1):
                if (a == 0 and b % 2 == 0) or (a == dd and b % 2 == 1) or (b == 0 and a % 2 == 1) or (b == dd and a % 2 == 0):
                    continue
                comp = 0 if (a + b) % 2 == 1 else 1
                self._where[(a, b)] = (comp, rank[comp])
                rank[comp] += 1

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.lib().dq_match_destroy(h)

    def tables(self, component):
        """(dist uint8 [n, n, 2], distB uint8 [n, 2], w10) of one component (0: X part, 1: Z part)."""
        n = self.nodes
        dist, distB, w = np.zeros((n, n, 2), np.uint8), np.zeros((n, 2), np.uint8), ctypes.c_int()
        self._lib.check(self._lib.lib().dq_match_get_tables(self._h, int(component), dist.ctypes.data_as(ctypes.c_void_p),
                                                           distB.ctypes.data_as(ctypes.c_void_p), ctypes.byref(w)))
        return dist, distB, w.value

    def decode(self, defects, both=None):
        """defects: integer array [batch, 2 components, 2 words] (bit i = i-th plaquette of the component) -> (class uint8 [batch],
        inexact uint8 [batch]) as torch CUDA tensors."""
        torch = self._torch
        dev = torch.as_tensor(np.ascontiguousarray(defects, dtype=np.uint64).view(np.int64)).cuda() if not torch.is_tensor(defects) else defects
        batch = dev.shape[0]
        cls = torch.empty(batch, dtype=torch.uint8, device="cuda")
        flag = torch.empty(batch, dtype=torch.uint8, device="cuda")
        both = (self.error_model != "X") if both is None else both
        self._lib.check(self._lib.lib().dq_match_decode(self._h, dev.data_ptr(), batch, int(both), cls.data_ptr(), flag.data_ptr(),
                                                       torch.cuda.current_stream().cuda_stream))
        return cls, flag

    def pack(self, grids):
        """[batch, d+1, d+1] syndrome grids -> defects uint64 [batch, 2, 2]."""
        grids = np.asarray(grids).reshape(-1, self.d + 1, self.d + 1)
        out = np.zeros((len(grids), 2, 2), dtype=np.uint64)
        for i, g in enumerate(grids):
            for (a, b) in zip(*np.nonzero(g)):
                comp, bit = self._where[(int(a), int(b))]
                out[i, comp, bit >> 6] |= np.uint64(1) << np.uint64(bit & 63)
        return out

    def predict(self, x, batch_size=None, verbose=0):
        cls, _ = self.decode(self.pack(x))
        out = np.zeros((len(cls), self.n_classes), dtype=np.float32)
        out[np.arange(len(cls)), cls.cpu().numpy()] = 1.0
        return out
