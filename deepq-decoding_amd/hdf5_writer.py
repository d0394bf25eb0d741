"""Minimal pure-Python writer for Keras 2.x HDF5 weight files (h5py is not available).

Writes the same on-disk structures the shipped agents use (/root/reference/trained_models/*/*/final_dqn_weights.h5f, as
dumped with hdf5_reader._File): superblock version 0 with 8-byte offsets / lengths, old-style groups (one v1 B-tree node
'TREE' -> one symbol-table node 'SNOD', names in a local heap 'HEAP'), version-1 object headers, and per dataset the
messages dataspace v1 (with max dims), datatype IEEE little-endian float32, fill value v2, layout v3 class 1 (contiguous),
modification time.  Keras' loader (`load_weights_from_hdf5_group`) needs three attributes, written as fixed-length
null-padded strings (what h5py produces for numpy bytes arrays): `layer_names` on the root group, `weight_names` on every
layer group, and `keras_version` (without it Keras assumes a Keras-1 file and transposes the kernels); `backend` is
written as well.  (The shipped files hold the two scalar attributes as variable-length strings in a global heap; Keras
only calls `.decode` on them, which fixed-length values support.)

Tree written:   /<layer>            group, attr weight_names = [<scope>/kernel:0, <scope>/bias:0]
                /<layer>/<scope>    group
                /<layer>/<scope>/kernel:0, bias:0   float32 datasets
where <scope> = <layer>, except that keras-rl's dueling rewrite names the last layer's variables `<layer>_1` (the shipped
files have /dense_3/dense_3_1/kernel:0).
"""
import struct
import time

import numpy as np

_UNDEF = 0xFFFFFFFFFFFFFFFF
_GROUP_LEAF_K, _GROUP_INTERNAL_K = 4, 16
_SNOD_SIZE = 8 + 2 * _GROUP_LEAF_K * 40
_TREE_SIZE = 24 + (2 * _GROUP_INTERNAL_K + 1) * 8 + 2 * _GROUP_INTERNAL_K * 8


def _pad8(b):
    return b + b"\x00" * (-len(b) % 8)


def _msg(mtype, data, flags=0):
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _object_header(messages):
    body = b"".join(messages)
    # version 1, reserved, number of messages, reference count 1, header size; the prefix is padded to 16 bytes
    return struct.pack("<BxHII4x", 1, len(messages), 1, len(body)) + body


def _string_type(size):
    # class 3 (string), version 1; bit field: null-padded, ASCII; size
    return struct.pack("<BBBBI", 0x13, 0x01, 0, 0, size)


def _attribute(name, values):
    """Attribute message (version 1) holding fixed-length strings: a 1-D array for a list, a scalar for a single bytes value."""
    scalar = isinstance(values, bytes)
    items = [values] if scalar else list(values)
    width = max(1, max(len(v) for v in items)) if items else 1
    dtype = _string_type(width)
    if scalar:
        space = struct.pack("<BBBB4x", 1, 0, 0, 0)                  # version 1, rank 0
    else:
        space = struct.pack("<BBBB4xQQ", 1, 1, 1, 0, len(items), len(items))      # rank 1, max dims present
    nm = name.encode() + b"\x00"
    data = b"".join(v.ljust(width, b"\x00") for v in items)
    body = struct.pack("<BxHHH", 1, len(nm), len(dtype), len(space)) + _pad8(nm) + _pad8(dtype) + _pad8(space) + data
    return _msg(0x0C, body)


class _Writer:
    def __init__(self):
        self.buf = bytearray(96)                                    # superblock + root symbol-table entry, filled in at the end

    def alloc(self, data):
        while len(self.buf) % 8:
            self.buf.append(0)
        addr = len(self.buf)
        self.buf += data
        return addr

    def dataset(self, arr, mtime):
        arr = np.ascontiguousarray(arr, dtype="<f4")
        raw = self.alloc(arr.tobytes())
        dims = b"".join(struct.pack("<Q", d) for d in arr.shape)
        msgs = [
            _msg(0x01, struct.pack("<BBBB4x", 1, arr.ndim, 1, 0) + dims + dims),                       # dataspace v1, dims + max dims
            _msg(0x03, bytes.fromhex("11201f000400000000002000170800177f000000"), flags=1),            # IEEE f32 little-endian
            _msg(0x05, bytes([2, 2, 2, 1]) + struct.pack("<I", 0), flags=1),                           # fill value v2: defined, size 0
            _msg(0x08, struct.pack("<BBQQ", 3, 1, raw, arr.nbytes)),                                   # layout v3, contiguous
            _msg(0x12, struct.pack("<B3xI", 1, mtime)),                                                # modification time
        ]
        return self.alloc(_object_header(msgs))

    def group(self, entries, attributes=()):
        """entries: {name: (object header address, is_group, btree, heap)}; returns (header, btree, heap) addresses."""
        assert len(entries) <= 2 * _GROUP_LEAF_K, "one symbol-table node per group"
        names = sorted(entries)                                     # symbol-table nodes are ordered by name
        heap_data = bytearray(8)                                    # offset 0: the empty string (B-tree key 0)
        offsets = {}
        for n in names:
            offsets[n] = len(heap_data)
            heap_data += _pad8(n.encode() + b"\x00")
        free_off = len(heap_data)
        heap_data += struct.pack("<QQ", 1, 16)                      # one free block at the end: next = 1 (none), size 16
        heap = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), free_off, 0))
        seg = self.alloc(bytes(heap_data))
        struct.pack_into("<Q", self.buf, heap + 24, seg)            # address of the data segment
        snod = bytearray(b"SNOD" + struct.pack("<BxH", 1, len(names)))
        for n in names:
            hdr, is_group, bt, hp = entries[n]
            snod += struct.pack("<QQI4x", offsets[n], hdr, 1 if is_group else 0)
            snod += struct.pack("<QQ", bt, hp) if is_group else bytes(16)
        snod += bytes(_SNOD_SIZE - len(snod))
        snod_addr = self.alloc(bytes(snod))
        tree = bytearray(b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if names else 0, _UNDEF, _UNDEF))
        tree += struct.pack("<Q", 0)                                # key 0: the empty string
        if names:
            tree += struct.pack("<QQ", snod_addr, offsets[names[-1]])      # child 0, key 1 = largest name in it
        tree += bytes(_TREE_SIZE - len(tree))
        tree_addr = self.alloc(bytes(tree))
        msgs = [_msg(0x11, struct.pack("<QQ", tree_addr, heap))] + list(attributes)
        return self.alloc(_object_header(msgs)), tree_addr, heap

    def finish(self, root):
        hdr, tree, heap = root
        while len(self.buf) % 8:
            self.buf.append(0)
        sb = b"\x89HDF\r\n\x1a\n" + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack("<HHI", _GROUP_LEAF_K, _GROUP_INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, _UNDEF, len(self.buf), _UNDEF)
        sb += struct.pack("<QQI4xQQ", 0, hdr, 1, tree, heap)        # root symbol-table entry (cached B-tree / heap addresses)
        assert len(sb) == 96
        self.buf[:96] = sb
        return bytes(self.buf)


def write_keras_weights(path, weights, layer_names, dueling_last=True, keras_version=b"2.2.2", backend=b"tensorflow"):
    """weights: [kernel, bias] per layer in Keras order (kernels HWIO / (in,out)); layer_names: conv2d_1.., dense_1.."""
    assert len(weights) == 2 * len(layer_names)
    w = _Writer()
    mtime = int(time.time())
    root_entries = {}
    for i, layer in enumerate(layer_names):
        scope = f"{layer}_1" if (dueling_last and i == len(layer_names) - 1) else layer
        k = w.dataset(weights[2 * i], mtime)
        b = w.dataset(weights[2 * i + 1], mtime)
        scope_group = w.group({"kernel:0": (k, False, 0, 0), "bias:0": (b, False, 0, 0)})
        names = [f"{scope}/kernel:0".encode(), f"{scope}/bias:0".encode()]
        layer_group = w.group({scope: (scope_group[0], True, scope_group[1], scope_group[2])}, [_attribute("weight_names", names)])
        root_entries[layer] = (layer_group[0], True, layer_group[1], layer_group[2])
    root = w.group(root_entries, [_attribute("layer_names", [n.encode() for n in layer_names]),
                                  _attribute("backend", backend), _attribute("keras_version", keras_version)])
    with open(path, "wb") as f:
        f.write(w.finish(root))
