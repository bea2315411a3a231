"""Scene files and rollouts -- the inference-side part of the reference's ``datasets/dataset_reader_physics.py``
(same function / class names where they exist there):

  Dataset            :179-207  a directory of ``*.msgpack.zst`` scene files, or in-memory scenes
  read_scene / write_scene     one scene file = one zstd frame holding a msgpack list of per-frame dicts whose
                               ndarrays use the msgpack-numpy encoding {nd, type, kind, shape, data}
                               (``load_data`` of run_sample.py:200-205, cache writer :168-176)
  get_rollout        :410-456  per-scene dicts of stacked frames (pos [T,N,3], vel, grav broadcast to [T,N,3], box ...)
                               with the translate / scale / grav_eqvar input transform of :276-293
  write_results      :520-526  HDF5 result file (needs h5py, which this image lacks: raises a clear error) and an
                               .npz stand-in with the same content

zstd comes from the system ``libzstd.so.1`` through ctypes (the ``zstandard`` wheel is not installed here); msgpack
from the ``msgpack`` package.  The training-side data flow (shuffling, augmentation, batching, the column / free-fall
generators) is out of scope (SURVEY.md section 8f rank 2 covers the file formats only).
"""
import ctypes
import ctypes.util
import glob
import os

import msgpack
import numpy as np

_zstd = None


def _libzstd():
    global _zstd
    if _zstd is None:
        name = ctypes.util.find_library("zstd") or "libzstd.so.1"
        z = ctypes.CDLL(name)
        z.ZSTD_getFrameContentSize.restype = ctypes.c_ulonglong
        z.ZSTD_getFrameContentSize.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        z.ZSTD_decompress.restype = ctypes.c_size_t
        z.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        z.ZSTD_compressBound.restype = ctypes.c_size_t
        z.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
        z.ZSTD_compress.restype = ctypes.c_size_t
        z.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        z.ZSTD_isError.restype = ctypes.c_uint
        z.ZSTD_isError.argtypes = [ctypes.c_size_t]
        _zstd = z
    return _zstd


def zstd_decompress(raw):
    z = _libzstd()
    n = z.ZSTD_getFrameContentSize(raw, len(raw))
    if n in (2 ** 64 - 1, 2 ** 64 - 2):  # ZSTD_CONTENTSIZE_UNKNOWN / _ERROR
        raise ValueError("not a zstd frame with a known content size")
    buf = ctypes.create_string_buffer(max(int(n), 1))
    r = z.ZSTD_decompress(buf, n, raw, len(raw))
    if z.ZSTD_isError(r):
        raise ValueError("zstd decompression failed")
    return buf.raw[:r]


def zstd_compress(data, level=19):
    z = _libzstd()
    cap = z.ZSTD_compressBound(len(data))
    buf = ctypes.create_string_buffer(cap)
    r = z.ZSTD_compress(buf, cap, data, len(data), level)
    if z.ZSTD_isError(r):
        raise ValueError("zstd compression failed")
    return buf.raw[:r]


def _decode(o):
    """msgpack-numpy object hook (keys may arrive as str or bytes)."""
    if isinstance(o, dict):
        nd = o.get("nd", o.get(b"nd"))
        if nd is not None and ("type" in o or b"type" in o):
            g = lambda k: o.get(k, o.get(k.encode()))  # noqa: E731
            dtype = g("type")
            dtype = np.dtype(dtype.decode() if isinstance(dtype, bytes) else dtype)
            if nd:
                return np.frombuffer(g("data"), dtype=dtype).reshape(g("shape")).copy()
            return np.frombuffer(g("data"), dtype=dtype)[0]
    return o


def _encode(o):
    """msgpack-numpy default hook: ndarrays and numpy scalars as {nd, type, kind, shape, data}."""
    if isinstance(o, np.ndarray):
        return {"nd": True, "type": o.dtype.str, "kind": "", "shape": list(o.shape), "data": np.ascontiguousarray(o).tobytes()}
    if isinstance(o, np.generic):
        return {"nd": False, "type": o.dtype.str, "data": o.tobytes()}
    raise TypeError(f"cannot serialise {type(o)}")


def read_scene(path):
    """One ``*.msgpack.zst`` file -> list of per-frame dicts (run_sample.py:200-205, Dataset.__getitem__ :199-207)."""
    with open(path, "rb") as f:
        return msgpack.unpackb(zstd_decompress(f.read()), raw=False, object_hook=_decode, strict_map_key=False)


def write_scene(path, frames, level=19):
    """Inverse of :func:`read_scene` (the cache writer of DatasetGroup.gen_data :168-176)."""
    with open(path, "wb") as f:
        f.write(zstd_compress(msgpack.packb(frames, use_bin_type=True, default=_encode), level))


class Dataset:
    """datasets/dataset_reader_physics.py:179-207."""

    def __init__(self, data=None, dataset_path=None):
        self.data, self.files = None, None
        if dataset_path is not None:
            self.files = sorted(glob.glob(os.path.join(dataset_path, "*.msgpack.zst")))
            assert len(self.files), "List of files must not be empty"
        elif data is not None:
            self.data = data
        else:
            raise NotImplementedError()

    def __len__(self):
        return len(self.data) if self.data is not None else len(self.files)

    def __getitem__(self, idx):
        return self.data[idx] if self.data is not None else read_scene(self.files[idx])


class DatasetGroup:
    """datasets/dataset_reader_physics.py:85-142, the part the test split needs: ``dataset_path`` holds the scene files
    (``*.msgpack.zst``) of the split, in ``<path>/test`` if that directory exists, else in ``<path>`` itself.  ``data``:
    scenes already in memory (a list of per-scene frame lists, e.g. a committed fixture).  The training-side generators
    (``type: column | free_fall`` without a dataset_path: the reference's own 1-D SPH solver, column_gen.py) are host code
    outside the per-step hot path and are not rebuilt here: pass ``data`` or a ``dataset_path``."""

    def __init__(self, train=None, valid=None, test=None, split="train", regen=False, data=None, **dataset_cfg):
        self.name = dataset_cfg.pop("name", "dataset")
        self.train = self.valid = self.test = None
        if data is not None:
            self.test = self.valid = Dataset(data=data)
            return
        if "dataset_path" not in dataset_cfg or dataset_cfg["dataset_path"] is None:
            raise NotImplementedError(
                f"dataset type {dataset_cfg.get('type', 'tank')!r} is generated by the reference's training-side solver "
                "(datasets/column_gen.py / free_fall_gen.py), which is outside the hot path: pass --dataset_path with "
                "scene files, or DatasetGroup(data=...)")
        path = dataset_cfg.pop("dataset_path")
        if split == "train":
            raise NotImplementedError("training is out of scope of the MI355X hot path (SURVEY.md section 2 row 16)")
        if split != "valid":  # :132-142
            sub = os.path.join(path, "test")
            self.test = Dataset(dataset_path=sub if os.path.exists(sub) else path)
            if split == "test":
                self.valid = self.test
        else:
            sub = os.path.join(path, "valid")
            self.valid = Dataset(dataset_path=sub if os.path.exists(sub) else path)


def align_vector(v0, v1):
    """Rotation taking v1 to v0 (:35-49), float32 like the reference's numpy code."""
    v0 = np.asarray(v0, dtype=np.float32)
    v1 = np.asarray(v1, dtype=np.float32)
    v0 = v0 / np.linalg.norm(v0)
    v1 = v1 / np.linalg.norm(v1)
    v = np.cross(v0, v1)
    c = np.dot(v0, v1)
    s = np.linalg.norm(v)
    if s < 1e-6:
        return (np.eye(3) * c).astype(np.float32)
    vx = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=np.float32)
    return (np.eye(3, dtype=np.float32) + vx + vx @ vx * ((1 - c) / (s * s))).astype(np.float32)


def get_rollout(dataset, stride=1, time_start=0, time_end=None, cnt=None, translate=None, scale=None, grav_eqvar=None,
                **kwargs):
    """:410-456 with PhysicsSimDataFlow(window=0) unrolled: one dict per scene with the selected frames stacked --
    ``pos / vel / grav [T,N,3]``, ``m / viscosity [T,N]``, ``frame_id / scene_id [T]``, ``box / box_normals [T,M,3]``
    (the static boundary of frame 0 repeated, :333-341) -- after the input transform of :276-293.  Scenes whose
    particle count changes over time cannot be stacked (as in the reference)."""
    out = []
    for si in range(len(dataset)):
        if cnt is not None and len(out) >= cnt:
            break
        frames = dataset[si]
        sel = [f for f in frames
               if int(f["frame_id"]) >= time_start * stride and int(f["frame_id"]) % stride == 0
               and (time_end is None or int(f["frame_id"]) < time_end * stride)]
        if not sel:
            continue
        merge = {}
        for k in ("pos", "vel", "grav", "m", "viscosity"):
            if k in sel[0] and sel[0][k] is not None:
                merge[k] = np.stack([np.asarray(f[k], dtype=np.float32) for f in sel], 0)
        for k in ("box", "box_normals"):
            b = np.asarray(frames[0][k], dtype=np.float32).reshape(-1, 3) if k in frames[0] else np.empty((0, 3), np.float32)
            merge[k] = np.stack([b for _ in sel], 0)
        merge["frame_id"] = np.asarray([f["frame_id"] for f in sel])
        merge["scene_id"] = np.asarray([f.get("scene_id", "") for f in sel])
        if "grav" in merge:
            merge["grav"] = np.broadcast_to(merge["grav"].reshape(len(sel), 1, 3), merge["vel"].shape).copy()  # :349-353
        if translate is not None:  # :276-293
            merge["pos"] = merge["pos"] + np.float32(translate)
            merge["box"] = merge["box"] + np.float32(translate)
        if scale is not None:
            for k in ("pos", "box", "vel", "grav"):
                if k in merge:
                    merge[k] = merge[k] * np.float32(scale)
        if grav_eqvar is not None and "grav" in merge:
            R = align_vector(grav_eqvar, merge["grav"][0, 0])
            merge["orig_grav"] = merge["grav"][0, 0].copy()
            for k in ("box", "box_normals", "pos", "vel", "grav"):
                merge[k] = np.matmul(merge[k], R)
        out.append(merge)
    return out


def write_results(path, name, data):
    """:520-526: HDF5 file with one group ``name`` and one dataset per (array, props) entry, attrs ``type`` / ``dim``
    -- what utils/draw_sim2d.py:170-174 reads.  Written through h5py when that is installed (the reference's own call
    sequence), otherwise by the built-in writer (dmcf_amd/utils/hdf5_writer.py: contiguous little-endian datasets in a
    version-0 HDF5 file; checked against the HDF5 C library in tests/test_hdf5_writer.py)."""
    try:
        import h5py
    except ImportError:
        h5py = None
    if h5py is not None:
        with h5py.File(os.path.join(path), "w") as f:
            grp = f.create_group(name)
            for d, props in data:
                dset = grp.create_dataset(props["name"], data=d)
                dset.attrs["type"] = props.get("type", "DENSITY")
                dset.attrs["dim"] = d.shape
        return
    from ..utils.hdf5_writer import write_hdf5
    write_hdf5(os.path.join(path), name,
               [(props["name"], np.asarray(d), {"type": props.get("type", "DENSITY"), "dim": np.asarray(np.shape(d), dtype=np.int64)})
                for d, props in data])


def write_results_npz(path, name, data):
    """Same content as :func:`write_results` in a numpy archive: ``<name>/<dataset>`` arrays and ``<name>/<dataset>.type``."""
    arrays = {}
    for d, props in data:
        arrays[f"{name}/{props['name']}"] = np.asarray(d)
        arrays[f"{name}/{props['name']}.type"] = np.asarray(props.get("type", "DENSITY"))
    np.savez_compressed(path, **arrays)
