"""A minimal HDF5 reader / writer for Picasso's localization files.

Picasso stores a localization table as ONE 1-D compound dataset ("locs",
"identifications", ...) in the root group of an HDF5 file written by h5py with its default
(earliest) library version: superblock version 0, version-1 object headers, a symbol-table
root group, contiguous layout, little-endian atomic members (picasso/io.py:2089-2110,
:2167-2188).  This module reads exactly that family of files and writes files of the same
structure, so that the GPU path can run file -> file on machines without h5py.  It is not a
general HDF5 implementation: chunked or compressed datasets, nested groups, variable-length
or array members raise NotImplementedError.

Layout written (all addresses absolute, 8-byte aligned):
    superblock v0 (96 B) | root object header | B-tree node | local heap + data | symbol node |
    one object header per dataset | raw records
"""
from __future__ import annotations

import struct

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K, INTERNAL_K = 4, 16


# ----------------------------------------------------------------------------------------
# datatype messages
# ----------------------------------------------------------------------------------------
def _pad8(b: bytes) -> bytes:
    return b + b"\0" * (-len(b) % 8)


def _atomic_type_message(dt: np.dtype) -> bytes:
    dt = np.dtype(dt)
    if dt.byteorder == ">":
        raise NotImplementedError("big-endian members are not written")
    if dt.kind == "f":
        if dt.itemsize == 4:
            bits, props = (0x20, 31, 0), struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
        elif dt.itemsize == 8:
            bits, props = (0x20, 63, 0), struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
        else:
            raise NotImplementedError(f"float{8 * dt.itemsize}")
        return struct.pack("<BBBBI", 0x11, bits[0], bits[1], bits[2], dt.itemsize) + props
    if dt.kind in "iu":
        flags = 0x08 if dt.kind == "i" else 0x00
        return struct.pack("<BBBBI", 0x10, flags, 0, 0, dt.itemsize) + struct.pack("<HH", 0, 8 * dt.itemsize)
    raise NotImplementedError(f"member dtype {dt}")


def _type_message(dt: np.dtype) -> bytes:
    dt = np.dtype(dt)
    if dt.names is None:
        return _atomic_type_message(dt)
    body = b""
    for name in dt.names:
        mdt, off = dt.fields[name][0], dt.fields[name][1]
        if mdt.names is not None or mdt.shape:
            raise NotImplementedError("nested / array members")
        body += _pad8(name.encode("utf-8") + b"\0")
        body += struct.pack("<IB3xI4x4I", off, 0, 0, 0, 0, 0, 0)      # offset, rank 0, permutation, four dimension sizes
        body += _atomic_type_message(mdt)
    n = len(dt.names)
    return struct.pack("<BBBBI", 0x16, n & 0xFF, (n >> 8) & 0xFF, 0, dt.itemsize) + body


def _parse_type(buf: bytes, pos: int):
    """-> (numpy dtype, end position) of the datatype message starting at buf[pos]."""
    cv, b0, b1, b2, size = struct.unpack_from("<BBBBI", buf, pos)
    cls, version = cv & 0x0F, cv >> 4
    pos += 8
    if cls == 0:                                       # fixed point
        order = ">" if b0 & 1 else "<"
        return np.dtype(f"{order}{'i' if b0 & 0x08 else 'u'}{size}"), pos + 4
    if cls == 1:                                       # floating point
        order = ">" if b0 & 1 else "<"
        return np.dtype(f"{order}f{size}"), pos + 12
    if cls == 6:                                       # compound
        n = b0 | (b1 << 8)
        names, formats, offsets = [], [], []
        for _ in range(n):
            end = buf.index(b"\0", pos)
            name = buf[pos:end].decode("utf-8")
            if version == 3:
                pos = end + 1
                nbytes = 1 if size < 256 else (2 if size < 65536 else (3 if size < (1 << 24) else 4))
                off = int.from_bytes(buf[pos:pos + nbytes], "little")
                pos += nbytes
            else:
                pos += (end + 1 - pos + 7) // 8 * 8
                off = struct.unpack_from("<I", buf, pos)[0]
                pos += 4
                if version == 1:
                    rank = buf[pos]
                    if rank:
                        raise NotImplementedError("array members")
                    pos += 1 + 3 + 4 + 4 + 16
            mdt, pos = _parse_type(buf, pos)
            names.append(name); formats.append(mdt); offsets.append(off)
        return np.dtype({"names": names, "formats": formats, "offsets": offsets, "itemsize": size}), pos
    raise NotImplementedError(f"HDF5 datatype class {cls}")


# ----------------------------------------------------------------------------------------
# writer
# ----------------------------------------------------------------------------------------
def _message(mtype: int, data: bytes, flags: int = 0) -> bytes:
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _object_header(messages: list[bytes]) -> bytes:
    body = b"".join(messages)
    return struct.pack("<BxHII4x", 1, len(messages), 1, len(body)) + body


def write(path: str, datasets: dict) -> None:
    """Write {name: 1-D numpy array (structured or atomic)} as datasets of the root group."""
    if not datasets or len(datasets) > 2 * LEAF_K:
        raise NotImplementedError(f"1..{2 * LEAF_K} datasets per file")
    names = sorted(datasets)                               # symbol table entries are kept in name order
    arrays = {k: np.ascontiguousarray(datasets[k]) for k in names}
    for k, a in arrays.items():
        if a.ndim != 1:
            raise NotImplementedError("only 1-D datasets")
    # local heap data: "" at 0, then the names, then one free block
    heap_data = bytearray(_pad8(b"\0"))
    name_off = {}
    for k in names:
        name_off[k] = len(heap_data)
        heap_data += _pad8(k.encode("utf-8") + b"\0")
    free_off = len(heap_data)
    heap_size = free_off + 32
    heap_data += struct.pack("<QQ", 1, heap_size - free_off) + b"\0" * (heap_size - free_off - 16)   # next = 1: end of list

    pos = 96
    root_header_addr = pos
    root_header_len = 16 + 8 + 16
    pos += root_header_len
    btree_addr = pos
    btree_len = 24 + (2 * INTERNAL_K + 1) * 8 + 2 * INTERNAL_K * 8
    pos += btree_len
    heap_addr = pos
    pos += 32
    heap_data_addr = pos
    pos += heap_size
    snod_addr = pos
    snod_len = 8 + 2 * LEAF_K * 40
    pos += snod_len

    headers, header_addr, data_addr = {}, {}, {}
    for k in names:                                         # object headers first (their size does not depend on addresses)
        a = arrays[k]
        msgs = [
            _message(0x0001, struct.pack("<BBB5xQ", 1, 1, 0, a.shape[0])),                       # dataspace v1, rank 1
            _message(0x0003, _type_message(a.dtype), flags=1),                                     # datatype (constant)
            _message(0x0005, struct.pack("<BBBB", 2, 2, 0, 0)),                                    # fill value v2: late alloc, undefined
            _message(0x0008, struct.pack("<BBQQ", 3, 1, 0, a.nbytes)),                             # layout v3 contiguous (address patched below)
        ]
        headers[k] = msgs
        header_addr[k] = pos
        pos += len(_object_header(msgs))
    for k in names:
        pos = (pos + 7) // 8 * 8
        data_addr[k] = pos if arrays[k].nbytes else UNDEF
        pos += arrays[k].nbytes
    eof = pos

    out = bytearray()
    out += SIGNATURE + struct.pack("<BBBBBBBxHHI", 0, 0, 0, 0, 0, 8, 8, LEAF_K, INTERNAL_K, 0)
    out += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    out += struct.pack("<QQI4xQQ", 0, root_header_addr, 1, btree_addr, heap_addr)                   # root symbol table entry
    assert len(out) == 96
    out += _object_header([_message(0x0011, struct.pack("<QQ", btree_addr, heap_addr))])
    assert len(out) == btree_addr
    node = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, snod_addr, name_off[names[-1]])
    out += node + b"\0" * (btree_len - len(node))
    out += b"HEAP" + struct.pack("<B3xQQQ", 0, heap_size, free_off, heap_data_addr)
    out += heap_data
    assert len(out) == snod_addr
    snod = b"SNOD" + struct.pack("<BxH", 1, len(names))
    for k in names:
        snod += struct.pack("<QQI4x16x", name_off[k], header_addr[k], 0)
    out += snod + b"\0" * (snod_len - len(snod))
    for k in names:
        a = arrays[k]
        msgs = headers[k]
        msgs[3] = _message(0x0008, struct.pack("<BBQQ", 3, 1, data_addr[k], a.nbytes))
        assert len(out) == header_addr[k]
        out += _object_header(msgs)
    for k in names:
        out += b"\0" * (-len(out) % 8)
        out += arrays[k].tobytes()
    assert len(out) == eof
    with open(path, "wb") as fh:
        fh.write(out)


# ----------------------------------------------------------------------------------------
# reader
# ----------------------------------------------------------------------------------------
class _File:
    def __init__(self, buf: bytes):
        self.buf = buf
        if buf[:8] != SIGNATURE:
            raise ValueError("not an HDF5 file")
        version = buf[8]
        if version > 1:
            raise NotImplementedError(f"HDF5 superblock version {version} (written with a non-default libver)")
        if buf[13] != 8 or buf[14] != 8:
            raise NotImplementedError("only 8-byte offsets and lengths")
        p = 24 + (4 if version == 1 else 0)
        self.base = struct.unpack_from("<Q", buf, p)[0]
        root = p + 32
        _, self.root_header, cache, self.root_btree, self.root_heap = struct.unpack_from("<QQI4xQQ", buf, root)
        if cache != 1:
            hdr = self.messages(self.root_header)
            st = [d for t, d in hdr if t == 0x0011]
            if not st:
                raise NotImplementedError("root group without a symbol table")
            self.root_btree, self.root_heap = struct.unpack_from("<QQ", st[0], 0)

    def messages(self, addr: int):
        """(type, data) of every message of a version-1 object header, following continuations."""
        buf = self.buf
        addr += self.base
        version, nmsg, _, size = struct.unpack_from("<BxHII", buf, addr)
        if version != 1:
            raise NotImplementedError(f"object header version {version}")
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            pos, length = blocks.pop(0)
            end = pos + length
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize = struct.unpack_from("<HH", buf, pos)
                data = buf[pos + 8:pos + 8 + msize]
                pos += 8 + msize
                if mtype == 0x0010:                         # continuation
                    off, ln = struct.unpack_from("<QQ", data, 0)
                    blocks.append((off + self.base, ln))
                out.append((mtype, data))
        return out

    def links(self):
        """{name: object header address} of the root group (symbol table B-tree)."""
        buf = self.buf
        heap = self.root_heap + self.base
        if buf[heap:heap + 4] != b"HEAP":
            raise ValueError("bad local heap")
        heap_data = struct.unpack_from("<Q", buf, heap + 24)[0] + self.base
        out = {}

        def walk(addr):
            addr += self.base
            if buf[addr:addr + 4] == b"TREE":
                _, level, used = struct.unpack_from("<BBH", buf, addr + 4)
                p = addr + 24
                for i in range(used):
                    child = struct.unpack_from("<Q", buf, p + 8 + i * 16)[0]
                    walk(child)
            elif buf[addr:addr + 4] == b"SNOD":
                n = struct.unpack_from("<H", buf, addr + 6)[0]
                for i in range(n):
                    noff, ohdr = struct.unpack_from("<QQ", buf, addr + 8 + i * 40)
                    s = heap_data + noff
                    out[buf[s:buf.index(b"\0", s)].decode("utf-8")] = ohdr
            else:
                raise ValueError("bad group node")

        walk(self.root_btree)
        return out

    def dataset(self, addr: int) -> np.ndarray:
        shape = dtype = None
        layout = None
        for mtype, data in self.messages(addr):
            if mtype == 0x0001:
                version, rank, flags = struct.unpack_from("<BBB", data, 0)
                p = 8 if version == 1 else 4
                shape = struct.unpack_from(f"<{rank}Q", data, p)
            elif mtype == 0x0003:
                dtype, _ = _parse_type(data, 0)
            elif mtype == 0x0008:
                version, cls = struct.unpack_from("<BB", data, 0)
                if version != 3:
                    raise NotImplementedError(f"data layout version {version}")
                if cls == 1:
                    layout = ("contiguous",) + struct.unpack_from("<QQ", data, 2)
                elif cls == 0:
                    n = struct.unpack_from("<H", data, 2)[0]
                    layout = ("compact", data[4:4 + n])
                else:
                    raise NotImplementedError("chunked datasets (compression / resizable) are not supported")
            elif mtype == 0x000B:
                raise NotImplementedError("filtered (compressed) datasets are not supported")
        if shape is None or dtype is None or layout is None:
            raise ValueError("incomplete dataset header")
        n = int(np.prod(shape)) if len(shape) else 1
        if layout[0] == "compact":
            raw = layout[1]
        elif layout[1] == UNDEF or n == 0:
            raw = b""
        else:
            raw = self.buf[layout[1] + self.base:layout[1] + self.base + layout[2]]
        return np.frombuffer(raw, dtype=dtype, count=n).reshape(shape).copy()


def read(path: str, name: str) -> np.ndarray:
    """The dataset `name` of the root group as a numpy (structured) array.  KeyError if absent."""
    with open(path, "rb") as fh:
        f = _File(fh.read())
    links = f.links()
    if name not in links:
        raise KeyError(f"No object named {name} in the file")
    return f.dataset(links[name])


def keys(path: str):
    with open(path, "rb") as fh:
        return sorted(_File(fh.read()).links())
