"""A minimal HDF5 writer: one group of contiguous little-endian datasets with small attributes -- the file
``datasets/dataset_reader_physics.py:520-526`` of the reference writes through h5py (group = model name; datasets ``pred`` /
``gt`` / ``bnd`` ``[T, N, 3]`` float32; attributes ``type`` (string) and ``dim`` (the shape)), which ``utils/draw_sim2d.py:170-174``
reads back.  h5py is not part of this image and the result writer must not depend on it.

What is written (HDF5 File Format Specification, the oldest and most widely readable forms): a version-0 superblock, version-1
object headers, "old style" groups (a symbol table message pointing at a one-leaf version-1 B-tree, a local heap with the link
names and one symbol table node), dataspace messages version 1, datatype messages version 1 (IEEE floats, two's-complement
integers, fixed-length strings), fill value message version 2, data layout message version 3 (contiguous), attribute messages
version 1.  No chunking, no compression, no free-space management: everything is laid out once, in one pass.
``tests/test_hdf5_writer.py`` reads the files back with the HDF5 C library itself (libhdf5 through ctypes, where the image has
it) and with an independent walk of the structures.
"""
import os
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K = 16       # symbol table nodes hold up to 2 * LEAF_K entries: one node per group
INTERNAL_K = 16   # B-tree nodes are allocated for 2 * INTERNAL_K children


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _datatype(dtype):
    """Datatype message (version 1) of a numpy dtype; strings: ``dtype`` = ('S', nbytes)."""
    if isinstance(dtype, tuple):  # fixed-length, null-terminated ASCII string
        return struct.pack("<BBBBI", 0x13, 0x00, 0, 0, dtype[1])
    dt = np.dtype(dtype)
    if dt.byteorder == ">":
        raise ValueError("big-endian arrays are not written")
    if dt.kind == "f" and dt.itemsize in (4, 8):
        exp_bits, man_bits = (8, 23) if dt.itemsize == 4 else (11, 52)
        bits = dt.itemsize * 8
        # class 1 (floating point), version 1; bit field: little-endian, mantissa normalisation 2 (implied leading bit) in bits 4-5,
        # sign bit location in the second byte
        head = struct.pack("<BBBBI", 0x11, 0x20, bits - 1, 0, dt.itemsize)
        props = struct.pack("<HHBBBBI", 0, bits, man_bits, exp_bits, 0, man_bits, (1 << (exp_bits - 1)) - 1)
        return head + props
    if dt.kind in "iu" and dt.itemsize in (1, 2, 4, 8):
        head = struct.pack("<BBBBI", 0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0, dt.itemsize)  # class 0, bit 3: signed
        return head + struct.pack("<HH", 0, dt.itemsize * 8)
    raise NotImplementedError(f"dtype {dt} (float32/64 and integers are written)")


def _dataspace(shape):
    """Dataspace message version 1 (rank 0: a scalar)."""
    return struct.pack("<BBBBI", 1, len(shape), 0, 0, 0) + b"".join(struct.pack("<Q", int(s)) for s in shape)


def _message(mtype, data, flags=0):
    data = _pad8(data)
    return struct.pack("<HHBBBB", mtype, len(data), flags, 0, 0, 0) + data


def _object_header(messages):
    body = b"".join(messages)
    # version, reserved, number of messages, reference count, header size; 4 bytes of padding bring the messages to offset 16
    return struct.pack("<BBHII", 1, 0, len(messages), 1, len(body)) + b"\0" * 4 + body


def _attribute(name, value):
    """Attribute message version 1: a str (fixed-length string, scalar) or an integer / float array (rank 0 or 1)."""
    nm = name.encode("ascii") + b"\0"
    if isinstance(value, (str, bytes)):
        raw = (value.encode("utf-8") if isinstance(value, str) else value) + b"\0"
        dt, ds, data = _datatype(("S", len(raw))), _dataspace(()), raw
    else:
        arr = np.ascontiguousarray(value)
        if arr.dtype.kind == "i" and arr.dtype.itemsize != 8:
            arr = arr.astype(np.int64)
        if arr.ndim > 1:
            raise NotImplementedError("attributes of rank > 1")
        dt, ds, data = _datatype(arr.dtype), _dataspace(arr.shape), arr.tobytes()
    head = struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(ds))
    return _message(0x000C, head + _pad8(nm) + _pad8(dt) + _pad8(ds) + data)


class _Layout:
    """The file, written as it is laid out: 8-byte aligned pieces appended straight to the (temporary) file -- a rollout's
    pred / gt arrays are tens of GB at 1M particles, and a byte image of the whole file in memory next to them (round 4) was
    three times that."""

    def __init__(self, f):
        self.f, self.size = f, 0

    def reserve(self, n):
        pad = -self.size % 8
        at = self.size + pad
        self.f.seek(self.size)
        self.f.write(b"\0" * (pad + n))
        self.size = at + n
        return at

    def put(self, at, b):
        self.f.seek(at)
        self.f.write(b)

    def add(self, b):
        """``b``: bytes or a C-contiguous numpy array (written from its own buffer, no copy)."""
        n = b.nbytes if isinstance(b, np.ndarray) else len(b)
        pad = -self.size % 8
        at = self.size + pad
        self.f.seek(self.size)
        if pad:
            self.f.write(b"\0" * pad)
        if n:
            self.f.write(memoryview(b).cast("B") if isinstance(b, np.ndarray) else b)
        self.size = at + n
        return at


def _group(lay, entries):
    """Write the structures of an old-style group whose members are ``entries`` = [(name, object header address, is_group,
    (btree, heap) | None)]; returns (object header address, btree address, heap address)."""
    if len(entries) > 2 * LEAF_K:
        raise NotImplementedError(f"more than {2 * LEAF_K} members in one group")
    entries = sorted(entries, key=lambda e: e[0].encode("ascii"))  # symbol table nodes are ordered by name (strcmp)
    # local heap data: the empty string at offset 0 (the first B-tree key), then the names, 8-byte aligned, null terminated
    heap_data = bytearray(b"\0" * 8)
    offs = []
    for name, *_ in entries:
        offs.append(len(heap_data))
        heap_data += _pad8(name.encode("ascii") + b"\0")
    seg = lay.add(bytes(heap_data))
    # local heap header: "HEAP", version 0, data segment size, head of the free list (1 = H5HL_FREE_NULL: none), data address
    heap = lay.add(b"HEAP" + struct.pack("<BBBBQQQ", 0, 0, 0, 0, len(heap_data), 1, seg))
    # symbol table node: "SNOD", version 1, number of symbols, 2 * LEAF_K entries of 40 bytes
    snod = bytearray(b"SNOD" + struct.pack("<BBH", 1, 0, len(entries)))
    for (name, addr, is_group, scratch), off in zip(entries, offs):
        snod += struct.pack("<QQII", off, addr, 1 if is_group else 0, 0)
        snod += struct.pack("<QQ", *scratch) if is_group else b"\0" * 16
    snod += b"\0" * (8 + 2 * LEAF_K * 40 - len(snod))
    snod_at = lay.add(bytes(snod))
    # B-tree node: "TREE", type 0 (group), level 0, one entry, no siblings; key 0 = heap offset 0 (""), child, key 1 = the
    # largest name of the child; allocated for 2 * INTERNAL_K children
    tree = bytearray(b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if entries else 0, UNDEF, UNDEF))
    tree += struct.pack("<QQQ", 0, snod_at, offs[-1] if offs else 0)
    tree += b"\0" * (24 + (2 * INTERNAL_K + 1) * 8 + 2 * INTERNAL_K * 8 - len(tree))
    btree = lay.add(bytes(tree))
    header = lay.add(_object_header([_message(0x0011, struct.pack("<QQ", btree, heap))]))
    return header, btree, heap


def write_hdf5(path, group, datasets):
    """``datasets``: [(name, array, {attribute name: str | array})] -> file ``path`` with ONE group ``group`` holding them.
    Written to ``path + ".tmp"`` and renamed when complete: a crash in the middle leaves no truncated .hdf5 behind."""
    tmp = path + ".tmp"
    try:
        with open(tmp, "wb") as f:
            lay = _Layout(f)
            lay.reserve(96)  # the superblock, written last (it holds the end-of-file address)
            members = []
            for name, arr, attrs in datasets:
                arr = np.ascontiguousarray(arr)
                if arr.dtype.kind == "f" and arr.dtype.itemsize not in (4, 8):
                    arr = arr.astype(np.float32)
                nbytes = arr.nbytes
                data_at = lay.add(arr) if nbytes else UNDEF
                msgs = [
                    _message(0x0001, _dataspace(arr.shape)),
                    _message(0x0003, _datatype(arr.dtype), flags=1),  # (constant message, as the library marks it)
                    _message(0x0005, struct.pack("<BBBB", 2, 2, 2, 0)),  # fill value v2: late allocation, written if set, undefined
                    _message(0x0008, struct.pack("<BBQQ", 3, 1, data_at, nbytes)),  # layout v3, contiguous
                ]
                msgs += [_attribute(k, v) for k, v in attrs.items()]
                members.append((name, lay.add(_object_header(msgs)), False, None))
            g_header, g_btree, g_heap = _group(lay, members)
            r_header, r_btree, r_heap = _group(lay, [(group, g_header, True, (g_btree, g_heap))])
            eof = lay.size + (-lay.size % 8)
            if eof > lay.size:
                lay.reserve(0)
            sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0)
            sb += struct.pack("<HHI", LEAF_K, INTERNAL_K, 0)
            sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
            sb += struct.pack("<QQII", 0, r_header, 1, 0) + struct.pack("<QQ", r_btree, r_heap)  # root group symbol table entry
            assert len(sb) == 96 and lay.size == eof
            lay.put(0, sb)
        os.replace(tmp, path)
    except BaseException:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise
