"""Feed-forward referee decoders with the reference's protocol.

The reference hands the environment a Keras model as ``static_decoder`` -- "a fast feed-forward NN homology class predictor"
(/root/reference/example_notebooks/Environments.py:53, loaded with ``load_model`` at
cluster_scripts/d5_dp/0.001/Single_Point_Training_Script.py:54-57) -- and calls
``static_decoder.predict(syndrome.reshape(1, (d+1)**2), batch_size=1, verbose=0)`` once per step, using only the argmax
(Environments.py:144,150).  The shipped blobs are missing (.MISSING_LARGE_BLOBS), so this module provides the same thing
from a weight file: a stack of Dense layers (ReLU between, softmax on top) read from a Keras HDF5 / .npz weight file.

Nothing here runs inside the environment step: ``VectorEnv.set_referee_predict`` tabulates ``predict`` once over all
2**n_stab syndromes into the kernel's joint look-up table, so on the device the referee is one table read whatever the
network's size.  The tabulation itself is a handful of small matrix products at construction time (numpy, on the host).
"""
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
