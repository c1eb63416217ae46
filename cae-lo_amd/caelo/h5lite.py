"""Minimal read-only HDF5 reader for Keras 2.2.x ``.h5`` model files.

The reference loads its two networks with ``keras.models.load_model(path)`` (Dirs.py:29-30,
Match.py:313,324).  Neither h5py nor Keras exists on the GPU box, so this module parses the
subset of HDF5 those files use: superblock v0, v1 object headers,
old-style groups (v1 B-tree + local heap + symbol-table nodes), contiguous little-endian
datasets, v1 attributes holding fixed- or variable-length strings (global heap).

API: ``H5File(path)`` -> ``.attrs(path)`` dict, ``.dataset(path)`` ndarray, ``.listdir(path)``.
"""
import struct

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(IOError):
    pass


class _Obj:
    __slots__ = ("msgs", "addr")

    def __init__(self, addr):
        self.addr = addr
        self.msgs = []  # (type, bytes)


class H5File:
    def __init__(self, path):
        with open(path, "rb") as f:
            self.buf = f.read()
        b = self.buf
        if b[:8] != _SIG:
            raise H5Error("%s: not an HDF5 file" % path)
        if b[8] != 0:
            raise H5Error("%s: only superblock version 0 is supported (got %d)" % (path, b[8]))
        self.O, self.L = b[13], b[14]
        if self.O != 8 or self.L != 8:
            raise H5Error("unsupported offset/length size %d/%d" % (self.O, self.L))
        self.base = struct.unpack_from("<Q", b, 24)[0]
        root_entry = 24 + 4 * 8
        self.root = self._sym_entry(root_entry)[1]
        self._gcol = {}

    # ---- low level ------------------------------------------------------------------
    def _u(self, fmt, off):
        return struct.unpack_from("<" + fmt, self.buf, off)

    def _sym_entry(self, off):
        name_off, hdr = self._u("QQ", off)
        return name_off, hdr + self.base

    def _object(self, addr):
        b = self.buf
        ver, _, nmsg, _refc, hsize = self._u("BBHII", addr)
        if ver != 1:
            raise H5Error("only v1 object headers are supported (got %d)" % ver)
        obj = _Obj(addr)
        blocks = [(addr + 16, hsize)]
        got = 0
        while blocks and got < nmsg:
            off, size = blocks.pop(0)
            end = off + size
            while off + 8 <= end and got < nmsg:
                mtype, msize, _flags = self._u("HHB", off)
                data = b[off + 8: off + 8 + msize]
                off += 8 + msize
                got += 1
                if mtype == 0x10:  # continuation
                    caddr, clen = struct.unpack_from("<QQ", data, 0)
                    blocks.append((caddr + self.base, clen))
                else:
                    obj.msgs.append((mtype, data))
        return obj

    def _heap_name(self, heap_addr, off):
        if self.buf[heap_addr:heap_addr + 4] != b"HEAP":
            raise H5Error("bad local heap")
        data_addr = self._u("Q", heap_addr + 8 + 2 * self.L)[0] + self.base
        s = data_addr + off
        e = self.buf.index(b"\0", s)
        return self.buf[s:e].decode("utf8")

    def _group_entries(self, obj):
        for t, d in obj.msgs:
            if t == 0x11:
                btree, heap = struct.unpack_from("<QQ", d, 0)
                out = {}
                self._walk_btree(btree + self.base, heap + self.base, out)
                return out
        raise H5Error("object at %d is not a group" % obj.addr)

    def _walk_btree(self, addr, heap, out):
        b = self.buf
        if b[addr:addr + 4] == b"SNOD":
            nsym = self._u("H", addr + 6)[0]
            for i in range(nsym):
                name_off, hdr = self._sym_entry(addr + 8 + 40 * i)
                out[self._heap_name(heap, name_off)] = hdr
            return
        if b[addr:addr + 4] != b"TREE":
            raise H5Error("bad B-tree node at %d" % addr)
        _ntype, _level, used = self._u("BBH", addr + 4)
        off = addr + 8 + 2 * self.O
        for i in range(used):
            child = self._u("Q", off + self.L + i * (self.L + self.O))[0]
            self._walk_btree(child + self.base, heap, out)

    def _resolve(self, path):
        addr = self.root
        for part in [p for p in path.split("/") if p]:
            ents = self._group_entries(self._object(addr))
            if part not in ents:
                raise KeyError(path)
            addr = ents[part]
        return self._object(addr)

    # ---- datatype / dataspace --------------------------------------------------------
    @staticmethod
    def _dataspace(d):
        ver, rank, flags = d[0], d[1], d[2]
        off = 8 if ver == 1 else 4
        return tuple(struct.unpack_from("<%dQ" % rank, d, off)) if rank else ()

    @staticmethod
    def _datatype(d):
        cls = d[0] & 0x0F
        bits0 = d[1]
        size = struct.unpack_from("<I", d, 4)[0]
        return cls, size, bits0

    def _global_heap_obj(self, caddr, index):
        if caddr not in self._gcol:
            b = self.buf
            if b[caddr:caddr + 4] != b"GCOL":
                raise H5Error("bad global heap collection")
            csize = self._u("Q", caddr + 8)[0]
            objs = {}
            off = caddr + 16
            while off + 16 <= caddr + csize:
                idx, _rc, _r, osize = self._u("HHIQ", off)
                if idx == 0:
                    break
                objs[idx] = b[off + 16: off + 16 + osize]
                off += 16 + ((osize + 7) // 8) * 8
            self._gcol[caddr] = objs
        return self._gcol[caddr][index]

    def _decode(self, dtype, shape, raw):
        cls, size, bits0 = dtype
        n = int(np.prod(shape)) if shape else 1
        if cls == 1:  # float
            if bits0 & 1:
                raise H5Error("big-endian floats not supported")
            return np.frombuffer(raw, dtype="<f%d" % size, count=n).reshape(shape).copy()
        if cls == 0:  # integer
            sign = "i" if (bits0 & 0x08) else "u"
            return np.frombuffer(raw, dtype="<%s%d" % (sign, size), count=n).reshape(shape).copy()
        if cls == 3:  # fixed string
            vals = [raw[i * size:(i + 1) * size].split(b"\0")[0] for i in range(n)]
            return vals[0] if shape == () else np.array(vals, dtype=object).reshape(shape)
        if cls == 9:  # variable length (string)
            vals = []
            for i in range(n):
                _ln, caddr, idx = struct.unpack_from("<IQI", raw, i * 16)
                vals.append(bytes(self._global_heap_obj(caddr + self.base, idx)))
            return vals[0] if shape == () else np.array(vals, dtype=object).reshape(shape)
        raise H5Error("unsupported datatype class %d" % cls)

    # ---- public ------------------------------------------------------------------------
    def listdir(self, path="/"):
        return sorted(self._group_entries(self._resolve(path)))

    def attrs(self, path="/"):
        out = {}
        for t, d in self._resolve(path).msgs:
            if t != 0x0C:
                continue
            ver = d[0]
            if ver != 1:
                raise H5Error("only v1 attribute messages are supported")
            nsz, tsz, ssz = struct.unpack_from("<HHH", d, 2)
            pad = lambda v: (v + 7) // 8 * 8
            off = 8
            name = d[off:off + nsz].split(b"\0")[0].decode("utf8")
            off += pad(nsz)
            dtype = self._datatype(d[off:off + tsz])
            off += pad(tsz)
            shape = self._dataspace(d[off:off + ssz])
            off += pad(ssz)
            out[name] = self._decode(dtype, shape, d[off:])
        return out

    def dataset(self, path):
        obj = self._resolve(path)
        shape = dtype = layout = None
        for t, d in obj.msgs:
            if t == 0x01:
                shape = self._dataspace(d)
            elif t == 0x03:
                dtype = self._datatype(d)
            elif t == 0x08:
                if d[0] != 3 or d[1] != 1:
                    raise H5Error("only contiguous (layout v3) datasets are supported")
                layout = struct.unpack_from("<QQ", d, 2)
        if shape is None or dtype is None or layout is None:
            raise H5Error("%s is not a dataset" % path)
        addr, size = layout
        if addr == _UNDEF:
            return np.zeros(shape, dtype="<f%d" % dtype[1])
        addr += self.base
        return self._decode(dtype, shape, self.buf[addr:addr + size])
