"""The built-in HDF5 writer (dmcf_amd/utils/hdf5_writer.py; the reference writes its results through h5py,
datasets/dataset_reader_physics.py:520-526) read back by the HDF5 C library itself -- libhdf5 through ctypes, where the image
ships it (conda's) -- and by an independent walk of the file's structures."""
import ctypes
import glob
import os
import struct

import numpy as np
import pytest

from dmcf_amd.utils.hdf5_writer import write_hdf5


def _libhdf5():
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib/libhdf5.so*"):
        for path in sorted(glob.glob(pat)):
            if "_hl" in path or "_cpp" in path or "fortran" in path:
                continue
            try:
                return ctypes.CDLL(path)
            except OSError:
                continue
    return None


def _read_with_libhdf5(L, path):
    """{group: {dataset: (array, {attribute: value})}} through the C API (hid_t is 64 bits from 1.10 on)."""
    c = ctypes
    hid = c.c_int64
    for name, res, args in (("H5Fopen", hid, [c.c_char_p, c.c_uint, hid]), ("H5Gopen2", hid, [hid, c.c_char_p, hid]),
                            ("H5Dopen2", hid, [hid, c.c_char_p, hid]), ("H5Dget_space", hid, [hid]), ("H5Dget_type", hid, [hid]),
                            ("H5Sget_simple_extent_ndims", c.c_int, [hid]), ("H5Sget_simple_extent_dims", c.c_int, [hid, c.c_void_p, c.c_void_p]),
                            ("H5Tget_size", c.c_size_t, [hid]), ("H5Tget_class", c.c_int, [hid]),
                            ("H5Dread", c.c_int, [hid, hid, hid, hid, hid, c.c_void_p]), ("H5Aopen", hid, [hid, c.c_char_p, hid]),
                            ("H5Aget_type", hid, [hid]), ("H5Aget_space", hid, [hid]), ("H5Aread", c.c_int, [hid, hid, c.c_void_p]),
                            ("H5Gget_num_objs", c.c_int, [hid, c.c_void_p]), ("H5Gget_objname_by_idx", c.c_ssize_t, [hid, c.c_uint64, c.c_char_p, c.c_size_t]),
                            ("H5Aget_num_attrs", c.c_int, [hid]), ("H5Aopen_idx", hid, [hid, c.c_uint]), ("H5Aget_name", c.c_ssize_t, [hid, c.c_size_t, c.c_char_p]),
                            ("H5Fclose", c.c_int, [hid]), ("H5open", c.c_int, [])):
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    assert L.H5open() >= 0
    f = L.H5Fopen(path.encode(), 0, 0)  # H5F_ACC_RDONLY, H5P_DEFAULT
    assert f >= 0, "libhdf5 cannot open the file"

    def members(g):
        n = c.c_uint64()
        assert L.H5Gget_num_objs(g, c.byref(n)) >= 0
        out = []
        for i in range(n.value):
            buf = c.create_string_buffer(256)
            L.H5Gget_objname_by_idx(g, i, buf, 256)
            out.append(buf.value.decode())
        return out

    def shape_of(space):
        nd = L.H5Sget_simple_extent_ndims(space)
        dims = (c.c_uint64 * max(nd, 1))()
        L.H5Sget_simple_extent_dims(space, dims, None)
        return tuple(int(d) for d in dims[:nd])

    def numpy_type(t):
        cls, size = L.H5Tget_class(t), L.H5Tget_size(t)
        return {(0, 4): np.int32, (0, 8): np.int64, (1, 4): np.float32, (1, 8): np.float64}.get((cls, size)), cls, size

    res = {}
    root = L.H5Gopen2(f, b"/", 0)
    for gname in members(root):
        g = L.H5Gopen2(f, gname.encode(), 0)
        assert g >= 0
        res[gname] = {}
        for dname in members(g):
            d = L.H5Dopen2(g, dname.encode(), 0)
            assert d >= 0
            t = L.H5Dget_type(d)
            nt, _, _ = numpy_type(t)
            arr = np.empty(shape_of(L.H5Dget_space(d)), dtype=nt)
            if arr.size:
                assert L.H5Dread(d, t, 0, 0, 0, arr.ctypes.data_as(c.c_void_p)) >= 0  # H5S_ALL, H5P_DEFAULT; file type = memory type
            attrs = {}
            for i in range(L.H5Aget_num_attrs(d)):
                a = L.H5Aopen_idx(d, i)
                nb = c.create_string_buffer(256)
                L.H5Aget_name(a, 256, nb)
                at = L.H5Aget_type(a)
                ant, cls, size = numpy_type(at)
                if cls == 3:  # string
                    buf = c.create_string_buffer(size)
                    assert L.H5Aread(a, at, buf) >= 0
                    attrs[nb.value.decode()] = buf.value.decode()
                else:
                    v = np.empty(shape_of(L.H5Aget_space(a)), dtype=ant)
                    assert L.H5Aread(a, at, v.ctypes.data_as(c.c_void_p)) >= 0
                    attrs[nb.value.decode()] = v
            res[gname][dname] = (arr, attrs)
    L.H5Fclose(f)
    return res


def _walk(path):
    """Independent reader of exactly what the writer documents: superblock -> root group -> the one group -> datasets."""
    b = open(path, "rb").read()
    assert b[:8] == b"\x89HDF\r\n\x1a\n" and b[8] == 0 and b[13] == 8 and b[14] == 8
    leaf_k, = struct.unpack_from("<H", b, 16)
    eof, = struct.unpack_from("<Q", b, 40)
    assert eof == len(b)
    root_hdr, = struct.unpack_from("<Q", b, 64)

    def messages(at):
        ver, _, n, _, size = struct.unpack_from("<BBHII", b, at)
        assert ver == 1
        p, out = at + 16, []
        for _ in range(n):
            t, sz, fl = struct.unpack_from("<HHB", b, p)
            out.append((t, b[p + 8:p + 8 + sz]))
            p += 8 + sz
        assert p == at + 16 + size
        return out

    def group_members(hdr):
        (t, data), = messages(hdr)
        assert t == 0x11
        btree, heap = struct.unpack("<QQ", data[:16])
        assert b[heap:heap + 4] == b"HEAP" and b[btree:btree + 4] == b"TREE"
        seg, = struct.unpack_from("<Q", b, heap + 24)
        used, = struct.unpack_from("<H", b, btree + 6)
        out = {}
        for e in range(used):
            snod, = struct.unpack_from("<Q", b, btree + 24 + 8 + 16 * e)
            assert b[snod:snod + 4] == b"SNOD"
            n, = struct.unpack_from("<H", b, snod + 6)
            assert n <= 2 * leaf_k
            for i in range(n):
                off, addr, cache = struct.unpack_from("<QQI", b, snod + 8 + 40 * i)
                name = b[seg + off:b.index(b"\0", seg + off)].decode()
                out[name] = addr
        return out

    res = {}
    for gname, ghdr in group_members(root_hdr).items():
        res[gname] = {}
        for dname, dhdr in group_members(ghdr).items():
            msg = dict((t, d) for t, d in messages(dhdr) if t != 0x0C)
            rank = msg[1][1]
            shape = struct.unpack_from("<%dQ" % rank, msg[1], 8)
            cls, size = msg[3][0] & 15, struct.unpack_from("<I", msg[3], 4)[0]
            dt = {(1, 4): np.float32, (1, 8): np.float64, (0, 4): np.int32, (0, 8): np.int64}[(cls, size)]
            assert msg[8][0] == 3 and msg[8][1] == 1
            addr, nbytes = struct.unpack_from("<QQ", msg[8], 2)
            arr = np.frombuffer(b[addr:addr + nbytes] if nbytes else b"", dtype=dt).reshape(shape)
            res[gname][dname] = arr
    return res


def _results(rng):
    pred = rng.normal(size=(7, 50, 3)).astype(np.float32)
    gt = rng.normal(size=(7, 50, 3)).astype(np.float32)
    bnd = rng.normal(size=(1, 20, 3)).astype(np.float32)
    return [(pred, {"name": "pred", "type": "PARTICLE"}), (gt, {"name": "gt", "type": "PARTICLE"}), (bnd, {"name": "bnd", "type": "PARTICLE"})]


def test_write_results_is_an_hdf5_file_the_c_library_reads(tmp_path):
    from dmcf_amd.datasets.dataset_reader_physics import write_results
    data = _results(np.random.default_rng(0))
    data.append((np.zeros((0, 3), np.float32), {"name": "empty"}))  # (utils/draw_sim2d.py:186 allows an empty boundary)
    path = str(tmp_path / "out.hdf5")
    write_results(path, "SymNet", data)
    walked = _walk(path)
    assert list(walked) == ["SymNet"] and sorted(walked["SymNet"]) == ["bnd", "empty", "gt", "pred"]
    for d, props in data:
        np.testing.assert_array_equal(walked["SymNet"][props["name"]], d)
    L = _libhdf5()
    if L is None:
        pytest.skip("no libhdf5 in this image: checked by the structural walk only")
    got = _read_with_libhdf5(L, path)
    assert list(got) == ["SymNet"] and sorted(got["SymNet"]) == ["bnd", "empty", "gt", "pred"]
    for d, props in data:
        arr, attrs = got["SymNet"][props["name"]]
        assert arr.dtype == np.float32
        np.testing.assert_array_equal(arr, d)
        assert attrs["type"] == props.get("type", "DENSITY")
        np.testing.assert_array_equal(attrs["dim"], np.asarray(d.shape, np.int64))


def test_other_dtypes_and_many_members(tmp_path):
    rng = np.random.default_rng(1)
    sets = [(f"d{i:02d}", rng.integers(-5, 5, size=(3, i + 1)).astype([np.int32, np.int64, np.float64, np.float32][i % 4]), {"k": np.float64(i)})
            for i in range(32)]
    path = str(tmp_path / "many.hdf5")
    write_hdf5(path, "g", sets)
    walked = _walk(path)["g"]
    for name, arr, _ in sets:
        np.testing.assert_array_equal(walked[name], arr)
    L = _libhdf5()
    if L is not None:
        got = _read_with_libhdf5(L, path)["g"]
        for name, arr, attrs in sets:
            assert got[name][0].dtype == arr.dtype
            np.testing.assert_array_equal(got[name][0], arr)
            assert float(np.asarray(got[name][1]["k"]).reshape(-1)[0]) == float(attrs["k"])
    with pytest.raises(NotImplementedError):
        write_hdf5(path, "g", sets + [("one_more", np.zeros(1, np.float32), {})])
