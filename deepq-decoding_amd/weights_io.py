"""Weight files.

* save: a Keras 2.x HDF5 weight file (hdf5_writer.py: the structures of the shipped agents, with the attributes Keras'
  `load_weights` reads) when the path ends in .h5 / .h5f / .hdf5 -- the reference saves `final_dqn_weights.h5f`
  (TRAIN:159-160) -- else a numpy .npz container holding the Keras-ordered tensors `<layer>/kernel:0`, `<layer>/bias:0`.
* load: either, told apart by the file's magic; HDF5 files as shipped under
  /root/reference/trained_models/*/*/final_dqn_weights.h5f are read by the minimal pure-Python reader in hdf5_reader.py
  (h5py is not available).
"""
import numpy as np

_HDF5_MAGIC = b"\x89HDF\r\n\x1a\n"


def save_weights_file(path, weights, layer_names, dueling=True):
    assert len(weights) == 2 * len(layer_names)
    if str(path).lower().endswith((".h5", ".h5f", ".hdf5")):
        from .hdf5_writer import write_keras_weights
        write_keras_weights(path, weights, layer_names, dueling_last=dueling)
        return
    arrays = {}
    for i, name in enumerate(layer_names):
        arrays[f"{name}/kernel:0"] = np.asarray(weights[2 * i], dtype=np.float32)
        arrays[f"{name}/bias:0"] = np.asarray(weights[2 * i + 1], dtype=np.float32)
    arrays["__order__"] = np.array(list(arrays.keys()))
    with open(path, "wb") as f:
        np.savez(f, **arrays)


def load_weights_file(path):
    with open(path, "rb") as f:
        magic = f.read(8)
    if magic == _HDF5_MAGIC:
        from .hdf5_reader import read_keras_weights
        return read_keras_weights(path)
    with np.load(path, allow_pickle=False) as z:
        return [z[k] for k in z["__order__"]]
