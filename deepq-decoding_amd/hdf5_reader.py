"""Minimal pure-Python reader for Keras 2.x HDF5 weight files (h5py is not available).

Handles exactly the subset such files use (checked on /root/reference/trained_models/*/*/final_dqn_weights.h5f):
superblock version 0, old-style groups (v1 B-tree 'TREE' + symbol-table nodes 'SNOD' + local heaps 'HEAP'),
version-1 object headers (with continuation blocks), contiguous little-endian float32 datasets (layout message
version 3, class 1).  No chunking, compression, fractal heaps or attributes are needed: datasets are found by walking
the group tree and ordered by Keras' layer naming (conv2d_1.., dense_1..; kernel before bias).
"""
import re
import struct

import numpy as np

_UNDEF = 0xFFFFFFFFFFFFFFFF


class _File:
    def __init__(self, path):
        with open(path, "rb") as f:
            self.b = f.read()
        if self.b[:8] != b"\x89HDF\r\n\x1a\n":
            raise ValueError("not an HDF5 file")
        if self.b[8] != 0:
            raise NotImplementedError(f"HDF5 superblock version {self.b[8]} is not supported (Keras 2.x files are version 0)")
        self.so, self.sl = self.b[13], self.b[14]
        if (self.so, self.sl) != (8, 8):
            raise NotImplementedError("only 8-byte offsets/lengths are supported")
        # superblock v0: 24 bytes of header fields, then base / free-space / EOF / driver addresses, then the root entry
        root_entry = 24 + 4 * 8
        self.root_header = self.u64(root_entry + 8)

    def u16(self, o): return struct.unpack_from("<H", self.b, o)[0]
    def u32(self, o): return struct.unpack_from("<I", self.b, o)[0]
    def u64(self, o): return struct.unpack_from("<Q", self.b, o)[0]

    # -- object headers --------------------------------------------------------------------------------------
    def messages(self, addr):
        """Yield (type, offset_of_data, size) for every message of the version-1 object header at addr."""
        if self.b[addr] != 1:
            raise NotImplementedError(f"object header version {self.b[addr]} is not supported")
        n_msgs, hdr_size = self.u16(addr + 2), self.u32(addr + 8)
        blocks = [(addr + 16, hdr_size)]
        seen = 0
        while blocks and seen < n_msgs:
            off, size = blocks.pop(0)
            end = off + size
            while off + 8 <= end and seen < n_msgs:
                mtype, msize = self.u16(off), self.u16(off + 2)
                data = off + 8
                seen += 1
                if mtype == 0x10:                                   # continuation
                    blocks.append((self.u64(data), self.u64(data + 8)))
                else:
                    yield mtype, data, msize
                off = data + msize

    # -- groups -------------------------------------------------------------------------------------------------
    def heap_string(self, heap_addr, offset):
        assert self.b[heap_addr:heap_addr + 4] == b"HEAP"
        data = self.u64(heap_addr + 24)
        end = self.b.index(b"\x00", data + offset)
        return self.b[data + offset:end].decode()

    def btree_entries(self, addr, heap):
        sig = self.b[addr:addr + 4]
        if sig == b"SNOD":
            n = self.u16(addr + 6)
            for i in range(n):
                e = addr + 8 + 40 * i
                yield self.heap_string(heap, self.u64(e)), self.u64(e + 8)
        elif sig == b"TREE":
            n = self.u16(addr + 6)
            base = addr + 8 + 16                                    # after node type/level/entries + two sibling addresses
            for i in range(n):
                child = self.u64(base + 8 + 16 * i)                  # key_i (8 bytes) then child_i (8 bytes)
                yield from self.btree_entries(child, heap)
        else:
            raise ValueError(f"unexpected node signature {sig!r}")

    def children(self, header_addr):
        for mtype, data, _ in self.messages(header_addr):
            if mtype == 0x11:                                       # symbol table message: b-tree + local heap
                yield from self.btree_entries(self.u64(data), self.u64(data + 8))

    def dataset(self, header_addr):
        """numpy array if the object is a contiguous float32 dataset, else None."""
        dims = addr = None
        is_f32 = False
        for mtype, data, _ in self.messages(header_addr):
            if mtype == 0x01:                                       # dataspace
                version, rank = self.b[data], self.b[data + 1]
                start = data + (8 if version == 1 else 4)
                dims = [self.u64(start + 8 * i) for i in range(rank)]
            elif mtype == 0x03:                                     # datatype: class in the low nibble of byte 0
                cls, size = self.b[data] & 0x0F, self.u32(data + 4)
                is_f32 = cls == 1 and size == 4 and (self.b[data + 1] & 1) == 0      # float, 4 bytes, little endian
            elif mtype == 0x08:                                     # layout
                if self.b[data] == 3 and self.b[data + 1] == 1:     # version 3, contiguous
                    addr = self.u64(data + 2)
        if dims is None or addr is None or addr == _UNDEF:
            return None
        if not is_f32:
            raise NotImplementedError("only little-endian float32 datasets are supported")
        n = int(np.prod(dims)) if dims else 1
        return np.frombuffer(self.b, dtype="<f4", count=n, offset=addr).reshape(dims).copy()

    def attributes(self, header_addr):
        """{name: list of bytes | bytes | None} for the attribute messages (version 1) of an object: fixed-length string arrays and
        scalars are decoded, anything else (the shipped files keep `backend` / `keras_version` as variable-length strings in a
        global heap) is reported as None."""
        out = {}
        for mtype, data, _ in self.messages(header_addr):
            if mtype != 0x0C or self.b[data] != 1:
                continue
            nsz, tsz, ssz = self.u16(data + 2), self.u16(data + 4), self.u16(data + 6)
            pad = lambda n: (n + 7) & ~7
            o = data + 8
            name = self.b[o:o + nsz].split(b"\x00")[0].decode()
            o += pad(nsz)
            cls, width = self.b[o] & 0x0F, self.u32(o + 4)
            o += pad(tsz)
            rank = self.b[o + 1]
            count = self.u64(o + 8) if rank == 1 else 1
            o += pad(ssz)
            if cls != 3 or rank > 1:
                out[name] = None
                continue
            vals = [self.b[o + i * width:o + (i + 1) * width].rstrip(b"\x00") for i in range(count)]
            out[name] = vals if rank == 1 else vals[0]
        return out

    def walk(self, header_addr=None, prefix=""):
        header_addr = self.root_header if header_addr is None else header_addr
        for name, child in self.children(header_addr):
            path = f"{prefix}/{name}"
            arr = self.dataset(child)
            if arr is not None:
                yield path, arr
            else:
                yield from self.walk(child, path)


def read_attributes(path):
    """{group path: {attribute: value}} for the root group and every group below it."""
    f = _File(path)
    out = {}

    def rec(addr, prefix):
        out[prefix or "/"] = f.attributes(addr)
        for name, child in f.children(addr):
            if f.dataset(child) is None:
                rec(child, f"{prefix}/{name}")
    rec(f.root_header, "")
    return out


def read_datasets(path):
    """{full path: float32 array} for every dataset in the file."""
    return dict(_File(path).walk())


_LAYER = re.compile(r"/(conv2d|dense)_(\d+)(?:_\d+)?/(kernel|bias):0$")


def read_keras_weights(path):
    """Weights in Keras model order: conv2d_1 kernel, bias, ..., dense_1 kernel, bias, ...  (kernels HWIO / (in,out))."""
    items = []
    for name, arr in read_datasets(path).items():
        m = _LAYER.search(name)
        if m:
            items.append(((0 if m.group(1) == "conv2d" else 1, int(m.group(2)), 0 if m.group(3) == "kernel" else 1), arr))
    if not items:
        raise ValueError("no conv2d_*/dense_* weights found")
    return [a for _, a in sorted(items, key=lambda x: x[0])]
