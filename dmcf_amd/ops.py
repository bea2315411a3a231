"""Operator surface of the hot path on MI355X -- the stand-in for ``open3d.ml.tf`` (``ml3d``).

Same names, argument meaning and error behaviour as the Open3D operators the reference calls
(paths relative to the reference tree):

  ml3d.layers.FixedRadiusSearch    utils/convolutions.py:207-210 (ctor), :354-358 (call),
                                   utils/tools/losses.py:296-298 (tuple-unpacked)
  ml3d.ops.continuous_conv         utils/convolutions.py:414-431, :454
  ml3d.ops.reduce_subarrays_sum    models/pbf_model.py:450-453

Tensors are torch CUDA (= ROCm) tensors; torch is only plumbing here (device memory, streams).
All arithmetic happens in libdmcf_hip.so through its C ABI (include/dmcf_hip.h).  There is no CPU
fallback: calling these with CPU tensors or without the built library raises.
"""
import collections
import os
import ctypes

import threading

import numpy as np
import torch

from . import _lib

class NeighborCapacityExceeded(RuntimeError):
    """A search enqueued with estimated buffer sizes produced more pairs than the buffers hold."""


class NeighborSearchResult:
    """(neighbors_index int32 [P], neighbors_row_splits int64 [m+1], neighbors_distance float32 [P]) as returned by
    ``ml3d.layers.FixedRadiusSearch`` -- attribute access and tuple unpacking both work
    (utils/convolutions.py:381-382, utils/tools/losses.py:297-298).

    The index / distance buffers may be LARGER than P when the search was enqueued without a host round trip
    (``capacity_hint``); the public attributes then synchronise once and return exact-length views.  Internal
    consumers use :meth:`raw`, which never synchronises (the kernels only need row_splits)."""

    def __init__(self, index_buf, row_splits, dist_buf, total=None, redo=None):
        self._index_buf, self._dist_buf = index_buf, dist_buf
        self.neighbors_row_splits = row_splits
        self._total = total
        self._redo = redo

    def raw(self):
        return self._index_buf, self.neighbors_row_splits, self._dist_buf

    def release(self):
        """Drop the index / distance buffers (the per-step neighbour cache calls this once a list's last consumer has
        enqueued its kernel: a 3e8-pair list is 2.4 GB, padded 3.4 GB).  Row splits / counts stay."""
        if self._index_buf is not None:
            self._capacity = self._index_buf.shape[0]  # (still needed to validate an estimated size at the end of the step)
        self._index_buf = self._dist_buf = None
        self._redo = None

    @property
    def total_ref(self):
        """0-dim device tensor holding P (no synchronisation)."""
        return self.neighbors_row_splits[-1]

    @property
    def capacity(self):
        return self._index_buf.shape[0] if self._index_buf is not None else self._capacity

    def resolve(self):
        """Make the result exact: one synchronisation; repeats the write pass if the estimate was too small."""
        if self._total is None:
            total = int(self.neighbors_row_splits[-1].item())
            if total > self._index_buf.shape[0]:
                self._index_buf, self._dist_buf = self._redo(pair_capacity(total))
            self._total = total
        return self._total

    def overflowed(self, total):
        return self._total is None and total > self.capacity

    @property
    def neighbors_index(self):
        return self._index_buf[:self.resolve()]

    @property
    def neighbors_distance(self):
        n = self.resolve()
        return self._dist_buf[:n] if self._dist_buf.shape[0] >= n else self._dist_buf

    def __iter__(self):
        return iter((self.neighbors_index, self.neighbors_row_splits, self.neighbors_distance))

    def __getitem__(self, i):
        return tuple(self)[i]

    def __len__(self):
        return 3

class PaddedNeighborList(NeighborSearchResult):
    """Result of the single-pass search (dmcf_frs_search_padded): row i lives at ``index[i * stride ...]`` with
    ``counts[i]`` entries; no count pass, no prefix scan, no host round trip.  ``raw()`` / ``row_count`` feed the
    kernels directly; the Open3D-style attributes compact the rows on first use (one synchronisation)."""

    def __init__(self, index_buf, row_begin, dist_buf, counts, max_count, stride, redo_exact):
        super().__init__(index_buf, row_begin, dist_buf, total=None, redo=None)
        self.row_count = counts          # int32 [m]
        self.max_count = max_count       # int32 [1] on the device: largest unclamped row
        self.stride = int(stride)
        self._redo_exact = redo_exact
        self._compact = None

    @property
    def total_ref(self):
        """0-dim device tensor holding P (no synchronisation); summed once per list, not once per consumer."""
        t = getattr(self, "_total_ref", None)
        if t is None:
            # (an int32 tensor summed into int64 costs a cast launch and the reduction; rows x stride bounds the total on the host)
            small = self.row_count.shape[0] * self.stride < 2 ** 31
            t = self._total_ref = self.row_count.sum(dtype=torch.int32) if small else self.row_count.sum()
        return t

    def release(self):
        super().release()
        self._redo_exact = None
        self._compact = None

    def overflowed(self, max_count):
        return max_count > self.stride

    def resolve(self):
        if self._compact is None:
            if int(self.max_count.item()) > self.stride:  # truncated rows: the exact two-pass search instead
                self._compact = self._redo_exact()
            else:
                m = self.row_count.shape[0]
                cnt = self.row_count.long()
                rs = torch.zeros(m + 1, dtype=torch.int64, device=cnt.device)
                torch.cumsum(cnt, 0, out=rs[1:])
                total = int(rs[-1].item())
                row = torch.repeat_interleave(torch.arange(m, device=cnt.device), cnt, output_size=total)
                src = row * self.stride + (torch.arange(total, device=cnt.device) - rs[:-1][row])
                dist = self._dist_buf[src] if self._dist_buf.shape[0] else self._dist_buf
                self._compact = NeighborSearchResult(self._index_buf[src], rs, dist, total=total)
        return self._compact

    @property
    def neighbors_index(self):
        return self.resolve().neighbors_index

    @property
    def neighbors_distance(self):
        return self.resolve().neighbors_distance

    @property
    def csr_row_splits(self):
        return self.resolve().neighbors_row_splits

    def __iter__(self):
        c = self.resolve()
        return iter((c.neighbors_index, c.neighbors_row_splits, c.neighbors_distance))


MAPPINGS = {"ball_to_cube_radial": 0, "ball_to_cube_volume_preserving": 1, "identity": 2}
INTERPOLATIONS = {"linear": 0, "linear_border": 1, "nearest_neighbor": 2}
WINDOWS = {None: 0, "explicit": 1, "poly6": 2, "cubic": 3, "linear": 4, "peak": 5, "cubic_grad": 6}

FLAG_ALIGN_CORNERS, FLAG_NORMALIZE, FLAG_SYMMETRIC, FLAG_ACCUMULATE, FLAG_SKIP_SELF, FLAG_FILTER_PACKED = 1, 2, 4, 8, 16, 32


class LaunchTimer:
    """Optional per-launch timing with HIP events on the stream the kernels are enqueued on (torch's current
    stream).  bench.py installs one to measure the CConv kernel's average launch duration inside the timed
    region; when ``ops.timer`` is None (the default) nothing is recorded."""

    def __init__(self):
        self.records = []  # (kind, meta dict, start event, end event)

    def begin(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        return ev

    def end(self, kind, meta, start):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        self.records.append((kind, meta, start, ev))

    def results(self):
        """-> list of (kind, meta, milliseconds); call after a device synchronise."""
        out = []
        for k, m, s, e in self.records:
            m = {a: (int(b.item()) if isinstance(b, torch.Tensor) else b) for a, b in m.items()}
            out.append((k, m, s.elapsed_time(e)))
        return out


timer = None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    # (the raw handle of torch's current stream: ~1 us; torch.cuda.current_stream() builds a Stream object, 12 us, and a
    # step of the 2-D models asks 80 times)
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev_f32(t, name, cols=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch tensor")
    if not t.is_cuda:
        raise _lib.DmcfError(f"{name} is on {t.device}: the DMCF hot path runs on the GPU only (no CPU fallback)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    if cols is not None and (t.dim() != 2 or t.shape[1] != cols):
        raise ValueError(f"{name} must have shape [n,{cols}], got {tuple(t.shape)}")
    t = t.contiguous()
    if t.data_ptr() % 16:  # the kernels use 16-byte vector loads on feature / filter rows
        t = t.clone()
    return t


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class SpatialHashTable:
    """Result of :func:`build_spatial_hash_table`: the cell-sorted grid of one point set for one radius.

    Plays the role of Open3D's (hash_table_index, hash_table_cell_splits, hash_table_splits) triple,
    which the reference can pass as ``fixed_radius_search_hash_table`` (utils/convolutions.py:283,358).
    """

    def __init__(self, points, radius, workspace, n_queries_capacity):
        self.points = points
        self.radius = float(radius)
        self.workspace = workspace
        self.n_queries_capacity = n_queries_capacity


def build_spatial_hash_table(points, radius, n_queries=None, **_ignored):
    """ml3d.ops.build_spatial_hash_table equivalent (hash_table_size_factor etc. are accepted and ignored:
    the structure is a dense cell-sorted grid, see dmcf_amd/csrc/frs.hip)."""
    L = _lib.lib()
    points = _dev_f32(points, "points", 3)
    n = points.shape[0]
    m = n if n_queries is None else int(n_queries)
    nbytes = L.dmcf_frs_workspace_bytes(n, m)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=points.device)
    t0 = timer.begin() if timer is not None else None
    _lib.check(L.dmcf_frs_build(_ptr(points), n, float(radius), _ptr(ws), nbytes, _stream()), "dmcf_frs_build")
    if timer is not None:
        timer.end("frs_build", dict(n_points=n), t0)
    return SpatialHashTable(points, radius, ws, m)


def reserve_device_memory(gib, device=None):
    """Make torch's caching allocator hold ONE free block of at least ``gib`` GiB (the allocator splits a cached block, it
    cannot join two segments): the multi-GB list buffers of a rollout -- and the bigger ones a scene grows into -- are then
    carved out of memory the process already owns instead of a fresh hipMalloc in the middle of a step (0.1 - 0.3 s each on
    an MI355X, with everything else queued behind it).  Optional; one GPU has 288 GB.  ``Simulator(reserve_gib=...)`` calls it.

    Returns the GiB this call took FROM THE DEVICE: ``gib`` when a new segment was created, 0.0 when the pool already held a
    free block that large (nothing to do -- the promise holds) and -1.0 when the device does not have that much to spare (the
    rollout then allocates as it goes)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    before = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    try:
        block = torch.empty(int(gib * (1 << 30)), dtype=torch.uint8, device=dev)
    except torch.cuda.OutOfMemoryError:
        return -1.0
    del block
    return float(gib) if torch.cuda.memory_stats(dev).get("num_device_alloc", 0) > before else 0.0


def pair_capacity(total):
    """Entries to allocate for a neighbour list of ``total`` pairs: 1/8 slack (the list of the next time step fits
    the same buffer) rounded up to 1/8 of the enclosing power of two, so that a rollout asks the caching allocator
    for the same few block sizes every step instead of a slightly different one each time (each new size is a
    hipMalloc of hundreds of MB: measured 12-18 GB of fresh allocations per step until sizes happened to repeat)."""
    x = int(total) + int(total) // 8 + 65536
    return _size_class(x)


def _size_class(x):
    """Buffer sizes (in 4-byte entries) the caching allocator can hand around: 1/8 of the enclosing power of two, and from
    512 MB on whole multiples of 1 GB -- the big lists of a step then fall into two or three classes, a block freed by
    one list fits the next one exactly instead of being split, and the pool stops growing after a step or two (with finer
    classes the pool of the 1M-particle rollout was still growing -- a fresh 3 GB hipMalloc, 100 ms -- in its fifth step)."""
    x = max(int(x), 1)
    if x >= 1 << 27:
        g = 1 << 28
    else:
        g = max(1 << 16, 1 << max(x.bit_length() - 4, 0))
    return (x + g - 1) // g * g


FRS_IGNORE_QUERY_POINT, FRS_OPEN3D_CORNER_VOXELS, FRS_OPEN3D_VOXEL_WALK = 1, 2, 4

# Which neighbour SET the searches return (include/dmcf_hip.h, DMCF_FRS_*):
#   "distance"        every point with d^2 <= R^2 -- what open3d's walk over the query's own voxel and the 8 corner voxels of
#                     q +- R covers in exact arithmetic.  Symmetric lists: the ASCC head conserves
#                     momentum (models/sym_net.py:42-53).  THE DEFAULT.
#   "open3d"          that walk as float arithmetic executes it: about one query in 10^6 (a rounding step from the middle of
#                     a voxel) keeps only what lies in its own voxel; bit-exact against oracle.fixed_radius_search(bins=
#                     "own+corners"), the oracle's default.
#   "open3d_corners"  round 3's reading of the library (the 8 corner voxels alone: such a query's row is nearly empty);
#                     bit-exact against bins="corners".
# The two emulations exist so that a capture of the real library (tools/capture_golden.py) can be matched bit for bit
# whichever way it falls; they cost a fixup pass per search and break the lists' symmetry exactly where the library does.
SEARCH_SETS = {"distance": 0, "open3d": FRS_OPEN3D_VOXEL_WALK, "open3d_corners": FRS_OPEN3D_CORNER_VOXELS}


_CONSTS = {}


def const_tensor(values, dtype, device):
    """A small constant tensor on the device, built once per (values, dtype, device): ``torch.tensor(list, device=...)`` is a
    blocking host -> device copy every time -- three of them per step were a third of the synchronising calls of a 2-D model's
    step (tools/profile_small.py).  The result is shared: do not write to it."""
    key = (tuple(float(v) for v in values), dtype, str(device))
    t = _CONSTS.get(key)
    if t is None:
        if len(_CONSTS) > 256:
            _CONSTS.clear()
        t = _CONSTS[key] = torch.tensor(list(values), dtype=dtype, device=device)
    return t


def search_set():
    """The neighbour set in force: environment variable DMCF_FRS_SET (see SEARCH_SETS), "distance" when unset."""
    if "DMCF_FRS_BRUTE_FORCE_SET" in os.environ:  # (round 3's switch; silently ignoring it would run another neighbour set than asked for)
        raise ValueError("DMCF_FRS_BRUTE_FORCE_SET is gone: use DMCF_FRS_SET=distance (what it selected) | open3d | open3d_corners")
    name = os.environ.get("DMCF_FRS_SET", "distance")
    if name not in SEARCH_SETS:
        raise ValueError(f"DMCF_FRS_SET={name!r}: expected one of {sorted(SEARCH_SETS)}")
    return name


def frs_flags(ignore_query_point):
    """Flags of the dmcf_frs_* entry points for the neighbour set in force (search_set)."""
    return (FRS_IGNORE_QUERY_POINT if ignore_query_point else 0) | SEARCH_SETS[search_set()]


def fixed_radius_search(points, queries, radius, ignore_query_point=False, return_distances=True,
                        hash_table=None, capacity_hint=None, row_stride=None, max_count=None):
    """-> NeighborSearchResult(neighbors_index int32 [P], neighbors_row_splits int64 [m+1],
    neighbors_distance float32 [P] (squared L2; empty if not return_distances)).

    ``capacity_hint``: an estimate of P.  With it count, scan and write are enqueued back to back with buffers of
    that size and NO host synchronisation; the result is validated later (see NeighborSearchResult).
    ``row_stride``: an upper bound of the row lengths.  With it ONE pass writes padded rows (PaddedNeighborList): no
    count pass at all; validated later through ``max_count``."""
    L = _lib.lib()
    points = _dev_f32(points, "points", 3)
    queries = _dev_f32(queries, "queries", 3)
    radius = float(radius)
    if not radius > 0:
        raise ValueError("radius must be positive")
    n, m = points.shape[0], queries.shape[0]
    if hash_table is None or hash_table.n_queries_capacity < m or hash_table.points.data_ptr() != points.data_ptr() \
            or hash_table.radius != radius:
        hash_table = build_spatial_hash_table(points, radius, n_queries=m)
    ws = hash_table.workspace
    nbytes = L.dmcf_frs_workspace_bytes(n, hash_table.n_queries_capacity)
    flags = frs_flags(ignore_query_point)
    dev = points.device
    row_splits = torch.empty(m + 1, dtype=torch.int64, device=dev)
    if row_stride is not None:
        stride = max(int(row_stride), 1)
        # allocation sizes in coarse buckets (see pair_capacity): the lattice point sets change size every step
        need = m * stride
        # 1/2 headroom (HBM is plentiful: the lists of the 1M-particle scene are 9 of 288 GB): the stride moves in steps of ~9 %
        # (row_stride) while a scene compresses or heats up, and a list that outgrows its buffer asks for a fresh multi-GB
        # block -- a hipMalloc of 25-150 ms behind a drained queue in that step; run to run the driver's 20-step window of
        # the bench scene cost 73 or 83 ms per step depending on how long its three reallocations happened to take
        cap = _size_class(need + need // 2)
        # a caller that repeats this search every step passes the capacity it got last time: while that still fits (and
        # is not grossly oversized) the request stays byte-identical, and the caching allocator answers it without a
        # hipMalloc (a fresh 2 GB block costs 20-120 ms; m * stride hovers around a bucket edge for steps on end)
        if capacity_hint is not None and need <= int(capacity_hint) <= 2 * cap:
            cap = int(capacity_hint)
        elif capacity_hint is not None and need > int(capacity_hint):
            # outgrown -- a scene that is compressing or heating up keeps growing: doubling makes this reallocation the last
            # one for a long while (the rollout of the 1M-particle box outgrew 1/4 headroom five times in 25 steps)
            cap = _size_class(2 * need)
        index = torch.empty(cap, dtype=torch.int32, device=dev)
        dist = torch.empty(cap if return_distances else 0, dtype=torch.float32, device=dev)
        counts = torch.empty(m, dtype=torch.int32, device=dev)
        if max_count is None:  # (a caller with many searches per step hands out slots of ONE zeroed tensor: a fill less per search)
            max_count = torch.zeros(1, dtype=torch.int32, device=dev)
        t0 = timer.begin() if timer is not None else None  # (after the allocations: a hipMalloc is not kernel time)
        _lib.check(L.dmcf_frs_search_padded(_ptr(queries), m, n, radius, flags, _ptr(ws), nbytes, stride, _ptr(row_splits),
                                            _ptr(counts), _ptr(index), _ptr(dist) if return_distances else None,
                                            _ptr(max_count), _stream()), "dmcf_frs_search_padded")
        keep = (points, queries, ws)  # noqa: F841
        res = PaddedNeighborList(index, row_splits, dist, counts, max_count, stride,
                                 lambda: fixed_radius_search(points, queries, radius, ignore_query_point, return_distances,
                                                             hash_table))
        if timer is not None:
            timer.end("frs_search_padded", dict(n_points=n, n_queries=m, pairs=res.total_ref, distances=bool(return_distances)), t0)
        return res
    t0 = timer.begin() if timer is not None else None
    _lib.check(L.dmcf_frs_count(_ptr(queries), m, n, radius, flags, _ptr(ws), nbytes, _ptr(row_splits), _stream()),
               "dmcf_frs_count")
    if timer is not None:
        timer.end("frs_count", dict(n_points=n, n_queries=m), t0)

    def write(capacity):
        index = torch.empty(capacity, dtype=torch.int32, device=dev)
        dist = torch.empty(capacity if return_distances else 0, dtype=torch.float32, device=dev)
        if capacity > 0 and m > 0:
            _lib.check(L.dmcf_frs_write(_ptr(queries), m, n, radius, flags, _ptr(ws), nbytes, _ptr(row_splits),
                                        _ptr(index), _ptr(dist) if return_distances else None, capacity, _stream()),
                       "dmcf_frs_write")
        return index, dist

    if capacity_hint is None:
        total = int(row_splits[-1].item())  # the one host round trip of the two-phase search
        t1 = timer.begin() if timer is not None else None
        index, dist = write(pair_capacity(total) if total else 0)
        res = NeighborSearchResult(index, row_splits, dist, total=total)
    else:
        t1 = timer.begin() if timer is not None else None
        index, dist = write(pair_capacity(capacity_hint))
        keep = (points, queries, ws)  # noqa: F841  (the closure keeps the operands alive for a possible redo)
        res = NeighborSearchResult(index, row_splits, dist, total=None, redo=write)
    if timer is not None:
        timer.end("frs_write", dict(n_points=n, n_queries=m, pairs=res.total_ref, distances=bool(return_distances)), t1)
    return res


class FixedRadiusSearch:
    """Mirror of ``ml3d.layers.FixedRadiusSearch`` (ctor utils/convolutions.py:207-210; call :354-358)."""

    def __init__(self, metric="L2", ignore_query_point=False, return_distances=False,
                 max_hash_table_size=32 * 2 ** 20, index_dtype=torch.int32, **kwargs):
        if metric != "L2":
            # every DMCF config uses the default radius_search_metric='L2' (utils/convolutions.py:165)
            raise NotImplementedError(f"metric {metric!r}: only 'L2' is implemented on the HIP path")
        if index_dtype != torch.int32:
            raise NotImplementedError("index_dtype must be int32 (Open3D 0.15.2 returns int32 indices)")
        self.metric = metric
        self.ignore_query_point = ignore_query_point
        self.return_distances = return_distances
        self.max_hash_table_size = max_hash_table_size

    def __call__(self, points, queries, radius, points_row_splits=None, queries_row_splits=None,
                 hash_table_size_factor=1 / 64, hash_table=None, capacity_hint=None, row_stride=None, max_count=None):
        if points_row_splits is not None or queries_row_splits is not None:
            raise NotImplementedError("batched row_splits are not used by DMCF (batch items are looped, "
                                      "pipelines/simulator.py:68-70)")
        if isinstance(radius, torch.Tensor):
            radius = float(radius)
        return fixed_radius_search(points, queries, radius, self.ignore_query_point, self.return_distances,
                                   hash_table=hash_table, capacity_hint=capacity_hint, row_stride=row_stride, max_count=max_count)

    call = __call__

    def index_only(self):
        """The same search without the distance output (for callers that re-form d^2 from the positions)."""
        twin = getattr(self, "_index_only", None)
        if twin is None:
            twin = self._index_only = FixedRadiusSearch(self.metric, self.ignore_query_point, False, self.max_hash_table_size)
        return twin


def _empty(t):
    return t is None or (isinstance(t, torch.Tensor) and t.numel() == 0)


def _cconv_args(filters, out_positions, extent, inp_positions, inp_features, neighbors_index, neighbors_row_splits,
                neighbors_value, window, window_fac, inp_importance, align_corners, coordinate_mapping, interpolation,
                normalize, symmetric, sym_axis, bias, out, accumulate, neighbors_row_count=None, filter_tile_mask=0,
                skip_self=False, row_length_hint=0):
    """Validate the operands and fill a ``dmcf_cconv_args``; returns (args, keepalive tensors, out)."""
    filters = _dev_f32(filters, "filters")
    if filters.dim() != 5:
        raise ValueError("filters must have shape [D,H,W,Cin,Cout]")
    out_positions = _dev_f32(out_positions, "out_positions", 3)
    inp_positions = _dev_f32(inp_positions, "inp_positions", 3)
    cin, cout = filters.shape[3], filters.shape[4]
    if inp_features is not None:
        inp_features = _dev_f32(inp_features, "inp_features", cin)
        if inp_features.shape[0] != inp_positions.shape[0]:
            raise ValueError("inp_features and inp_positions disagree on the number of points")
    n_out = out_positions.shape[0]
    if neighbors_index.dtype != torch.int32 or neighbors_row_splits.dtype != torch.int64:
        raise TypeError("neighbors_index must be int32 and neighbors_row_splits int64")
    if neighbors_row_splits.shape[0] != n_out + 1:
        raise ValueError("neighbors_row_splits must have n_out+1 entries")
    if window not in WINDOWS:
        raise NotImplementedError(f"window {window!r}")
    if window is not None:
        if _empty(neighbors_value):
            if window == "explicit" and neighbors_index.numel() > 0:
                raise ValueError("explicit window but no per-neighbour values")
            neighbors_value = None  # distance windows: the kernel re-forms d^2 from the positions (as the search does)
        else:
            neighbors_value = _dev_f32(neighbors_value, "neighbors_value")
            if neighbors_value.shape[0] != neighbors_index.shape[0]:
                raise ValueError("neighbors_value and neighbors_index disagree on the number of pairs")
    inp_importance = None if _empty(inp_importance) else _dev_f32(inp_importance, "inp_importance")
    if bias is not None:
        bias = _dev_f32(bias, "bias")
    neighbors_index = neighbors_index.contiguous()
    neighbors_row_splits = neighbors_row_splits.contiguous()
    a = _lib.CconvArgs()
    a.filters = filters.data_ptr()
    for d in range(5):
        a.filter_dims[d] = filters.shape[d]
    a.sym_axis = int(sym_axis)
    a.out_positions = out_positions.data_ptr()
    a.n_out = n_out
    a.inp_positions = inp_positions.data_ptr()
    a.n_inp = inp_positions.shape[0]
    a.inp_features = None if inp_features is None else inp_features.data_ptr()
    a.inp_importance = None if inp_importance is None else inp_importance.data_ptr()
    a.neighbors_index = neighbors_index.data_ptr()
    a.neighbors_row_splits = neighbors_row_splits.data_ptr()
    a.neighbors_value = None if window is None or neighbors_value is None else neighbors_value.data_ptr()
    a.extent = float(extent)
    a.window_fac = float(window_fac)
    a.window = WINDOWS[window]
    a.coordinate_mapping = MAPPINGS[coordinate_mapping]
    a.interpolation = INTERPOLATIONS[interpolation]
    a.flags = ((FLAG_ALIGN_CORNERS if align_corners else 0) | (FLAG_NORMALIZE if normalize else 0) |
               (FLAG_SYMMETRIC if symmetric else 0) | (FLAG_ACCUMULATE if accumulate else 0) | (FLAG_SKIP_SELF if skip_self else 0))
    a.bias = None if bias is None else bias.data_ptr()
    a.out = None if out is None else out.data_ptr()
    a.n_pairs = neighbors_index.shape[0]
    a.neighbors_row_count = None
    a.filter_tile_mask = int(filter_tile_mask) & 0xffffffff
    a.row_length_hint = int(row_length_hint)
    if neighbors_row_count is not None:
        if neighbors_row_count.dtype != torch.int32 or neighbors_row_count.shape[0] != n_out:
            raise TypeError("neighbors_row_count must be int32 [n_out]")
        neighbors_row_count = neighbors_row_count.contiguous()
        a.neighbors_row_count = neighbors_row_count.data_ptr()
    keep = (neighbors_row_count, filters, out_positions, inp_positions, inp_features, inp_importance, neighbors_index, neighbors_row_splits,
            neighbors_value, bias, out)
    return a, keep


_STENCILS = {}


def lattice_offsets(voxel, radius, device, shift=(0.0, 0.0, 0.0)):
    """int32 [S, 4] device tensor: the integer offsets d (x, y, z, 0) of input cells with |d * voxel - shift| <= radius,
    decided like the search decides a pair (fp32, un-fused squared distance), ordered z, y, x."""
    key = (tuple(float(v) for v in voxel), float(radius), tuple(float(v) for v in shift), str(device))
    st = _STENCILS.get(key)
    if st is None:
        v, sh = np.asarray(voxel, np.float32), np.asarray(shift, np.float32)
        reach = [int(np.floor((radius + abs(float(c))) / float(x))) + 1 if x > 0 else 0 for x, c in zip(v, sh)]
        ax = [np.arange(-r, r + 1, dtype=np.int32) for r in reach]
        dz, dy, dx = np.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
        x, y, z = dx.astype(np.float32) * v[0] - sh[0], dy.astype(np.float32) * v[1] - sh[1], dz.astype(np.float32) * v[2] - sh[2]
        d2 = (x * x + y * y) + z * z
        keep = d2 <= np.float32(radius) * np.float32(radius)
        off = np.stack([dx[keep], dy[keep], dz[keep], np.zeros(int(keep.sum()), np.int32)], axis=1).astype(np.int32)
        rch = [int(np.abs(off[:, k]).max()) if off.shape[0] else 0 for k in range(3)]
        st = (torch.from_numpy(np.ascontiguousarray(off)).to(device), rch)
        _STENCILS[key] = st
    return st[0]


def lattice_reach(voxel, radius, device, shift=(0.0, 0.0, 0.0)):
    """max |d| per axis (x, y, z) over :func:`lattice_offsets`: how far around the cells it covers a launch reads the input
    volume (the volume handed to :func:`lattice_conv` must be padded with zero cells that far)."""
    lattice_offsets(voxel, radius, device, shift)
    return list(_STENCILS[(tuple(float(v) for v in voxel), float(radius), tuple(float(v) for v in shift), str(device))][1])


def lattice_volume_box(base_min, base_dims, inp_step, reach, points_min=None, points_dims=None):
    """(min, dims) (x, y, z) of the box of input cells a launch of :func:`lattice_conv` over the base box can touch -- its x
    extent rounded up to whole 16-cell tiles -- united with the box of the input points."""
    lo, hi = [], []
    for k in range(3):
        ext = (int(base_dims[k]) + 15) // 16 * 16 if k == 0 else int(base_dims[k])
        l = int(base_min[k]) * inp_step - int(reach[k])
        h = (int(base_min[k]) + ext - 1) * inp_step + int(reach[k])
        if points_min is not None:
            l, h = min(l, int(points_min[k])), max(h, int(points_min[k]) + int(points_dims[k]) - 1)
        lo.append(l)
        hi.append(h)
    return lo, [hi[k] - lo[k] + 1 for k in range(3)]


def _lattice_args(filters, inp_volume, inp_min, out_table, out_min, n_out, voxel, extent, inp_step, out_stride, out_phase,
                  rel_shift, base_min, base_dims, window, window_fac, align_corners, coordinate_mapping, interpolation, bias,
                  out, accumulate):
    """dmcf_lattice_conv_args for one launch; returns (args, tensors to keep alive, number of stencil offsets)."""
    dev = filters.device
    offsets = lattice_offsets(voxel, 0.5 * float(extent), dev, rel_shift)
    if base_min is None:
        base_min, base_dims = out_min, [int(out_table.shape[2 - k]) for k in range(3)]
    a = _lib.LatticeConvArgs()
    a.filters = _ptr(filters)
    for k in range(5):
        a.filter_dims[k] = int(filters.shape[k])
    a.inp_volume, a.out_table = _ptr(inp_volume), _ptr(out_table)
    for k in range(3):
        a.inp_min[k], a.inp_dims[k] = int(inp_min[k]), int(inp_volume.shape[2 - k])
        a.out_min[k], a.out_dims[k] = int(out_min[k]), int(out_table.shape[2 - k])
        a.out_phase[k], a.base_min[k], a.base_dims[k] = int(out_phase[k]), int(base_min[k]), int(base_dims[k])
        a.rel_shift[k], a.voxel[k] = float(rel_shift[k]), float(voxel[k])
    a.n_out, a.inp_step, a.out_stride = int(n_out), int(inp_step), int(out_stride)
    a.offsets, a.n_offsets = _ptr(offsets), int(offsets.shape[0])
    for k, r in enumerate(lattice_reach(voxel, 0.5 * float(extent), dev, rel_shift)):
        a.reach[k] = r
    a.extent, a.window_fac = float(extent), float(window_fac)
    a.window = WINDOWS[window]
    a.coordinate_mapping, a.interpolation = MAPPINGS[coordinate_mapping], INTERPOLATIONS[interpolation]
    a.flags = (FLAG_ALIGN_CORNERS if align_corners else 0) | (FLAG_ACCUMULATE if accumulate else 0)
    a.bias = _ptr(bias) if bias is not None else None
    a.out = _ptr(out)
    return a, (offsets,), int(offsets.shape[0])


def lattice_conv(filters, inp_volume, inp_min, out_table, out_min, n_out, voxel, extent, inp_step=1, out_stride=1,
                 out_phase=(0, 0, 0), rel_shift=(0.0, 0.0, 0.0), base_min=None, base_dims=None, window="poly6",
                 window_fac=1.0, align_corners=True, coordinate_mapping="ball_to_cube_volume_preserving",
                 interpolation="linear", bias=None, out=None, accumulate=False, fill=1.0, n_out_launch=None, parts=None):
    """dmcf_lattice_conv_forward: continuous_conv between two aligned regular lattices without a neighbour list.
    ``inp_volume`` float32 [dz, dy, dx, Cin]: the input features by cell (zeros where no point is), entry 0 = input cell
    ``inp_min`` (x, y, z), padded so that it holds every cell ``a * inp_step + d`` of the launch (:func:`lattice_volume_box`); ``out_table`` int32 [dz, dy, dx]: output point index per cell of the output lattice (-1: none),
    entry 0 = output cell ``out_min``; ``voxel`` the input lattice spacing (x, y, z).  The launch covers the output cells
    ``a * out_stride + out_phase`` for the base vectors a of the box (``base_min``, ``base_dims``; default: the whole output
    table with stride 1); their stencil is ``a * inp_step + d`` (see include/dmcf_hip.h).
    ``parts``: a list of up to 8 dicts (out_phase, rel_shift, base_min, base_dims) that replace those four arguments and
    run as ONE grid (dmcf_lattice_conv_forward_batch); every part writes rows of its own."""
    L = _lib.lib()
    dev = filters.device
    cin, cout = filters.shape[3], filters.shape[4]
    if out is None:
        if accumulate:
            raise ValueError("accumulate=True needs an out tensor")
        out = torch.zeros((n_out, cout), dtype=torch.float32, device=dev)  # rows without a cell stay 0
    filters = filters.contiguous()
    if inp_volume.dim() != 4 or inp_volume.shape[3] != cin or not inp_volume.is_contiguous() or not out_table.is_contiguous():
        raise ValueError("inp_volume must be a contiguous [dz, dy, dx, Cin] tensor, out_table a contiguous [dz, dy, dx] one")
    if parts is None:
        parts = [dict(out_phase=out_phase, rel_shift=rel_shift, base_min=base_min, base_dims=base_dims)]
    arr = (_lib.LatticeConvArgs * len(parts))()
    keep, n_off = [], 0
    for i, pt in enumerate(parts):
        arr[i], k, no = _lattice_args(filters, inp_volume, inp_min, out_table, out_min, n_out, voxel, extent, inp_step, out_stride,
                                      pt["out_phase"], pt["rel_shift"], pt["base_min"], pt["base_dims"], window, window_fac,
                                      align_corners, coordinate_mapping, interpolation, bias, out, accumulate)
        keep.append(k)
        n_off += no
    if len(parts) == 1:
        nbytes = L.dmcf_lattice_conv_workspace_bytes(arr)
    else:
        nbytes = L.dmcf_lattice_conv_batch_workspace_bytes(arr, len(parts))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    t0 = timer.begin() if timer is not None else None
    if len(parts) == 1:
        _lib.check(L.dmcf_lattice_conv_forward(arr, _ptr(ws), nbytes, _stream()), "dmcf_lattice_conv_forward")
    else:
        _lib.check(L.dmcf_lattice_conv_forward_batch(arr, len(parts), _ptr(ws), nbytes, _stream()), "dmcf_lattice_conv_forward_batch")
    if timer is not None:
        # bench accounting: this form reads no neighbour list, so it is charged what it does read and write -- the input
        # volume once, the per-offset matrices, the cell -> point table and the output rows (``pairs_equiv`` = the pairs the
        # neighbour-list form would have had, for information only: outputs x stencil offsets x the fraction of occupied
        # cells of the input lattice's box)
        no = int(n_out if n_out_launch is None else n_out_launch)
        timer.end("cconv", dict(pairs=0, pairs_equiv=int(no * (n_off / len(parts)) * float(fill)), n_out=no, cin=int(cin),
                                cout=int(cout), K=int(filters.shape[0] * filters.shape[1] * filters.shape[2]), symmetric=False,
                                lattice=True, kernel="lat_conv_kernel", n_offsets=int(n_off), parts=len(parts),
                                volume_bytes=int(inp_volume.numel()) * 4, table_bytes=int(out_table.numel()) * 4,
                                accumulate=bool(accumulate)), t0)
    return out


def block_diagonal_tile_mask(blocks):
    """``filter_tile_mask`` of include/dmcf_hip.h for a filter whose only non-zero entries lie in the given blocks
    ``[(c0, c1, o0, o1), ...]`` (input channels c0 .. c1 - 1 into output channels o0 .. o1 - 1): bit 4 * (c / 4) + o / 16 for
    every (c, o) of a block.  0 (no hint) when the filter does not fit the mask (c >= 32 or o >= 64)."""
    mask = 0
    for c0, c1, o0, o1 in blocks:
        if c1 > 32 or o1 > 64:
            return 0
        for q in range(c0 // 4, (c1 + 3) // 4):
            for n in range(o0 // 16, (o1 + 15) // 16):
                mask |= 1 << (4 * q + n)
    return mask


def cconv_forward(filters, out_positions, extent, inp_positions, inp_features, neighbors_index,
                  neighbors_row_splits, neighbors_value=None, window=None, window_fac=1.0, inp_importance=None,
                  align_corners=True, coordinate_mapping="ball_to_cube_volume_preserving", interpolation="linear",
                  normalize=False, symmetric=False, sym_axis=2, bias=None, out=None, accumulate=False,
                  n_pairs_ref=None, neighbors_row_count=None, filter_tile_mask=0, skip_self=False, name_only=False,
                  row_length_hint=0, packed_cache=None):
    """One call of dmcf_cconv_forward.  ``row_length_hint``: 0 unknown / 1 tens / 2 hundreds of neighbours per row -- what the
    caller knows about the LAYER from its configuration (include/dmcf_hip.h).  ``skip_self``: DMCF_FLAG_SKIP_SELF (the list holds the query points, the layer ignores them; only the direct kernel).  ``name_only``: no launch, returns the name of the kernel these arguments dispatch to.  ``filter_tile_mask``: see ``block_diagonal_tile_mask`` (0 = no hint).  ``neighbors_row_count``: int32 [n_out] for padded lists (PaddedNeighborList).  ``window``: None | 'explicit' (neighbors_value = importance) |
    'poly6' | 'cubic' | 'linear' | 'peak' | 'cubic_grad' (neighbors_value = squared distances).
    """
    L = _lib.lib()
    n_out, cout = out_positions.shape[0], filters.shape[4]
    if out is None:
        if accumulate:
            raise ValueError("accumulate=True needs an out tensor")
        out = torch.empty((n_out, cout), dtype=torch.float32, device=filters.device)
    elif out.shape != (n_out, cout) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("out has the wrong shape / dtype / layout")
    if inp_positions.shape[0] == 0 or n_out == 0:
        # an empty input set (e.g. every boundary particle cropped away, pbf_model.py:330-336): every row is empty, the
        # result is the bias; nothing to launch (the C ABI rejects NULL point arrays)
        b = 0.0 if bias is None else bias.to(torch.float32)
        if accumulate:
            out += b
        else:
            out[:] = b
        return out
    a, keep = _cconv_args(filters, out_positions, extent, inp_positions, inp_features, neighbors_index,
                          neighbors_row_splits, neighbors_value, window, window_fac, inp_importance, align_corners,
                          coordinate_mapping, interpolation, normalize, symmetric, sym_axis, bias, out, accumulate,
                          neighbors_row_count, filter_tile_mask, skip_self, row_length_hint)
    if name_only:
        name = ctypes.create_string_buffer(96)
        _lib.check(L.dmcf_cconv_kernel_name(ctypes.byref(a), name, 96), "dmcf_cconv_kernel_name")
        return name.value.decode()
    nbytes = L.dmcf_cconv_workspace_bytes(ctypes.byref(a))
    ws = None
    if packed_cache is not None:
        # ``packed_cache``: a dict the calling LAYER owns.  It keeps the workspace of the layer's last call; while the filter
        # tensor (storage, version), its interpretation and the kernel the dispatch picks are the same, the packed filter in it
        # is still valid and is not formed again (DMCF_FLAG_FILTER_PACKED: one launch less per layer and step)
        name = ctypes.create_string_buffer(96)
        L.dmcf_cconv_kernel_name(ctypes.byref(a), name, 96)
        # Identity of the filter VALUES: the tensor object itself (held by the cache, so neither its address nor its id can be
        # reused by another tensor while the entry lives) + its version counter.  In-place writes through autograd-visible
        # calls (copy_, load_state_dict, optimiser steps) bump the counter; a write through ``.data`` does not -- after one,
        # call ``layer.invalidate_packed()``.  Derived tensors (a fresh object per call) never hit.
        key = (filters._version, tuple(filters.shape), bool(symmetric), int(sym_axis), name.value, str(filters.device))
        ws = packed_cache.get("ws")
        if packed_cache.get("src") is filters and packed_cache.get("key") == key and ws is not None and ws.numel() >= nbytes:
            a.flags |= FLAG_FILTER_PACKED
        else:
            # (the workspace of a small launch also holds scratch that grows with n_out: a quarter of headroom, so that a scene
            # whose point counts drift does not repack every step)
            ws = torch.empty(nbytes + nbytes // 4, dtype=torch.uint8, device=filters.device)
            packed_cache["key"], packed_cache["ws"], packed_cache["src"] = key, ws, filters
        nbytes = ws.numel()
    if ws is None:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=filters.device)
    t0 = timer.begin() if timer is not None else None
    _lib.check(L.dmcf_cconv_forward(ctypes.byref(a), _ptr(ws), nbytes, _stream()), "dmcf_cconv_forward")
    if timer is not None:
        kdims = [int(d) for d in filters.shape[:3]]
        if symmetric:
            kdims[int(sym_axis)] *= 2
        name = ctypes.create_string_buffer(96)
        L.dmcf_cconv_kernel_name(ctypes.byref(a), name, 96)
        timer.end("cconv", dict(pairs=n_pairs_ref if n_pairs_ref is not None else int(a.n_pairs), n_out=n_out, cin=int(filters.shape[3]), cout=cout,
                                K=kdims[0] * kdims[1] * kdims[2], symmetric=bool(symmetric), kernel=name.value.decode(),
                                pair_values=bool(a.neighbors_value), accumulate=bool(accumulate)), t0)
    return out


class ScatterPlan:
    """dmcf_cconv_scatter_plan's output (include/dmcf_hip.h): the input points counting-sorted by the block of ``block_cells``^3
    lattice cells they lie in.  Depends on the two point sets, the lattice spacing and the radius only -- one plan serves every
    layer of a step between the same two sets.  Built on the device without a host round trip."""

    def __init__(self, buf, voxel, block_cells, reach, n_inp, keep):
        self.buf, self.voxel, self.block_cells, self.reach, self.n_inp = buf, float(voxel), int(block_cells), int(reach), int(n_inp)
        self._keep = keep  # (the operands the plan was made from)


SCATTER_BLOCK_CELLS = 4  # m: blocks of m^3 lattice cells (0.4 units of the 0.1 lattice of Liquid3d: ~500 particles)


def scatter_reach(radius, voxel):
    return int(np.ceil(np.float32(radius) / np.float32(voxel) - 1e-4))


def scatter_plan(inp_positions, out_positions, voxel, radius, block_cells=None):
    L = _lib.lib()
    inp = _dev_f32(inp_positions, "inp_positions", 3)
    out = _dev_f32(out_positions, "out_positions", 3)
    m = int(block_cells or SCATTER_BLOCK_CELLS)
    n = inp.shape[0]
    nbytes = L.dmcf_cconv_scatter_plan_bytes(n)
    buf = torch.empty(nbytes, dtype=torch.uint8, device=inp.device)
    t0 = timer.begin() if timer is not None else None
    _lib.check(L.dmcf_cconv_scatter_plan(_ptr(inp), n, _ptr(out), out.shape[0], float(voxel), 2.0 * float(radius), m, _ptr(buf), nbytes,
                                         _stream()), "dmcf_cconv_scatter_plan")
    if timer is not None:
        timer.end("scatter_plan", dict(n_points=n), t0)
    return ScatterPlan(buf, voxel, m, scatter_reach(radius, voxel), n, (inp, out))


def scatter_kernel_name(cout, block_cells, reach):
    """The instantiation dmcf_cconv_scatter_forward launches, as rocprofv3 prints it (csrc/cconv_sct.hip: sct_waves)."""
    ns = (block_cells + 2 * reach + 1) ** 3
    ns = (ns + 3) & ~3

    def lds(waves):
        return cout * ns * 8 + ns * 4 + 2 * 2 * waves * 64 * cout * 4 + 16
    waves = 8 if (2 * lds(8) <= 160 * 1024 or cout != 4 or lds(16) > 160 * 1024) else 16
    return f"cconv_sct_kernel<{cout}, {waves}>"


def cconv_scatter_supported(filters, block_cells, reach):
    """Does dmcf_cconv_scatter_forward take a layer of this shape (4x4x4 filter, 4 or 8 outputs, a box that fits the LDS)?"""
    if tuple(filters.shape[:3]) != (4, 4, 4) or filters.shape[3] > 32 or filters.shape[4] not in (4, 8):
        return False
    return block_cells + 2 * reach + 1 <= (13 if filters.shape[4] == 4 else 11)


def cconv_scatter_forward(filters, out_positions, extent, inp_positions, inp_features, t_index, t_row_begin, t_row_count, plan,
                          window=None, window_fac=1.0, bias=None, out=None, accumulate=False, error_flag=None, n_pairs_ref=None):
    """One call of dmcf_cconv_scatter_forward (splat S: filter first, input stationary, 64-bit fixed-point sums): the operator
    of cconv_forward for particles -> coarse lattice layers with 4 or 8 output channels, walking the TRANSPOSED list (row j = the
    output points within extent / 2 of input point j; ``t_row_count`` None for CSR row splits)."""
    L = _lib.lib()
    filters = _dev_f32(filters, "filters")
    out_positions = _dev_f32(out_positions, "out_positions", 3)
    inp_positions = _dev_f32(inp_positions, "inp_positions", 3)
    cin, cout = int(filters.shape[3]), int(filters.shape[4])
    inp_features = _dev_f32(inp_features, "inp_features", cin)
    n_out, n_inp = out_positions.shape[0], inp_positions.shape[0]
    if plan.n_inp != n_inp:
        raise ValueError("the plan was made for another input point set")
    if window not in (None, "poly6"):
        raise NotImplementedError("cconv_scatter_forward: window must be None or 'poly6'")
    if out is None:
        if accumulate:
            raise ValueError("accumulate=True needs an out tensor")
        out = torch.empty((n_out, cout), dtype=torch.float32, device=filters.device)
    elif out.shape != (n_out, cout) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("out has the wrong shape / dtype / layout")
    a = _lib.CconvScatterArgs()
    a.filters = _ptr(filters)
    for k in range(5):
        a.filter_dims[k] = int(filters.shape[k])
    a.out_positions, a.n_out = _ptr(out_positions), n_out
    a.inp_positions, a.n_inp = _ptr(inp_positions), n_inp
    a.inp_features = _ptr(inp_features)
    a.t_index, a.t_row_begin = _ptr(t_index), _ptr(t_row_begin)
    a.t_row_count = _ptr(t_row_count) if t_row_count is not None else None
    a.t_capacity = int(t_index.shape[0])
    a.plan = _ptr(plan.buf)
    a.block_cells, a.reach = plan.block_cells, plan.reach
    a.extent, a.window_fac, a.window = float(extent), float(window_fac), WINDOWS[window]
    a.flags = FLAG_ALIGN_CORNERS | (FLAG_ACCUMULATE if accumulate else 0)
    a.bias = _ptr(bias) if bias is not None else None
    a.out = _ptr(out)
    a.error_flag = _ptr(error_flag) if error_flag is not None else None
    nbytes = L.dmcf_cconv_scatter_workspace_bytes(ctypes.byref(a))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=filters.device)
    t0 = timer.begin() if timer is not None else None
    _lib.check(L.dmcf_cconv_scatter_forward(ctypes.byref(a), _ptr(ws), nbytes, _stream()), "dmcf_cconv_scatter_forward")
    if timer is not None:
        pairs = n_pairs_ref if n_pairs_ref is not None else (t_row_count.sum() if t_row_count is not None else t_row_begin[-1])
        timer.end("cconv", dict(pairs=pairs, n_out=n_out, cin=cin, cout=cout, K=64, symmetric=False,
                                kernel=scatter_kernel_name(cout, plan.block_cells, plan.reach), pair_values=False, accumulate=bool(accumulate)), t0)
    return out


def continuous_conv(filters, out_positions, extents, offset, inp_positions, inp_features, inp_importance,
                    neighbors_index, neighbors_row_splits, neighbors_importance, align_corners=True,
                    coordinate_mapping="ball_to_cube_radial", interpolation="linear", normalize=True,
                    max_temp_mem_MB=64, **_ignored):
    """Mirror of ``ml3d.ops.continuous_conv`` with the keyword set of utils/convolutions.py:414-431.

    ``extents`` must hold a single value (DMCF always passes a scalar extent, :352-353,:390-392);
    ``offset`` must be zero (:200-201).  An empty tensor means "absent" for the importance inputs (:335,:376).
    """
    ext = extents if isinstance(extents, torch.Tensor) else torch.as_tensor(extents)
    if ext.numel() != 1:
        raise NotImplementedError("per-point extents (RadiusSearch path, utils/convolutions.py:366-370) "
                                  "are not used by DMCF and not implemented")
    if offset is not None and bool(torch.as_tensor(offset).ne(0).any()):
        raise NotImplementedError("non-zero offset is not used by DMCF and not implemented")
    window = None if _empty(neighbors_importance) else "explicit"
    return cconv_forward(filters, out_positions, float(ext), inp_positions, inp_features, neighbors_index,
                         neighbors_row_splits, neighbors_value=neighbors_importance, window=window,
                         inp_importance=inp_importance, align_corners=align_corners,
                         coordinate_mapping=coordinate_mapping, interpolation=interpolation, normalize=normalize)


def dense_supported(x, kernel):
    """Does :func:`dense_forward` take this product (else the caller keeps torch's GEMM)?"""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous() and x.shape[1] % 4 == 0
            and 0 < x.shape[1] <= 64 and 0 < kernel.shape[1] <= 64 and x.data_ptr() % 16 == 0)


def dense_forward(x, kernel, bias=None, residual=None):
    """``x @ kernel (+ bias) (+ residual)``: the networks' Dense layers on a million rows (dmcf_dense_forward)."""
    L = _lib.lib()
    n, k = x.shape
    m = kernel.shape[1]
    kernel = _dev_f32(kernel, "kernel")
    out = torch.empty(n, m, dtype=torch.float32, device=x.device)
    if residual is not None:
        residual = _dev_f32(residual, "residual")
        if tuple(residual.shape) != (n, m):
            raise ValueError("residual must be [n, m]")
    _lib.check(L.dmcf_dense_forward(_ptr(x), n, k, _ptr(kernel), m, _ptr(bias) if bias is not None else None,
                                    _ptr(residual) if residual is not None else None, _ptr(out), _stream()), "dmcf_dense_forward")
    return out


def points_aabb(points):
    """(min, max) over axis 0 of [n, 3] float32 points, two [3] device tensors: the fluid bounds of the boundary crop
    (models/pbf_model.py:330-336) in two small launches (a torch reduction over the strided columns was 0.2 ms + a transpose)."""
    L = _lib.lib()
    points = _dev_f32(points, "points")
    if points.dim() != 2 or points.shape[1] != 3:
        raise ValueError("points must be [n, 3]")
    out = torch.empty(6, dtype=torch.float32, device=points.device)
    wsb = int(L.dmcf_points_aabb_workspace_bytes())
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=points.device)
    _lib.check(L.dmcf_points_aabb(_ptr(points), points.shape[0], _ptr(out), _ptr(ws), wsb, _stream()), "dmcf_points_aabb")
    return out[:3], out[3:]


def reduce_subarrays_sum(values, row_splits):
    """Mirror of ``o3dml.ops.reduce_subarrays_sum`` (models/pbf_model.py:450-453)."""
    L = _lib.lib()
    values = _dev_f32(values, "values")
    if row_splits.dtype != torch.int64:
        raise TypeError("row_splits must be int64")
    n_rows = row_splits.shape[0] - 1
    out = torch.empty(n_rows, dtype=torch.float32, device=values.device)
    _lib.check(L.dmcf_reduce_subarrays_sum(_ptr(values), _ptr(row_splits.contiguous()), n_rows, _ptr(out), _stream()),
               "dmcf_reduce_subarrays_sum")
    return out


def neighbor_counts(row_splits):
    """``reduce_subarrays_sum(ones_like(neighbors_index), row_splits)`` without materialising the ones
    (models/pbf_model.py:450-453): float32 neighbour count per row.  Also accepts a search result."""
    if isinstance(row_splits, PaddedNeighborList):
        return row_splits.row_count.to(torch.float32)
    if isinstance(row_splits, NeighborSearchResult):
        row_splits = row_splits.neighbors_row_splits
    L = _lib.lib()
    if row_splits.dtype != torch.int64 or not row_splits.is_cuda:
        raise _lib.DmcfError("row_splits must be an int64 GPU tensor")
    n_rows = row_splits.shape[0] - 1
    out = torch.empty(n_rows, dtype=torch.float32, device=row_splits.device)
    _lib.check(L.dmcf_reduce_subarrays_sum(None, _ptr(row_splits.contiguous()), n_rows, _ptr(out), _stream()),
               "dmcf_reduce_subarrays_sum")
    return out


GRID_MAX_CELLS = 1 << 31  # dense cell table of the lattice bounding box: 4 bytes per cell, at most 8 GiB ...
# ... and at most this many cells per particle (+ a floor): a few particles that left the scene (a splash; a particle falling
# forever) stretch the bounding box, and a table that grows with it is a fresh multi-GB hipMalloc in every step of the
# rollout (measured on the 100k-particle dam break: 6 -> 8 GB per step until the 8 GiB cap).  Beyond it: the sort-based form.
GRID_MAX_CELLS_PER_POINT = 64
GRID_MIN_CELLS = 1 << 22


_GRID_CELLS = {}  # (voxel sizes of the levels, flags) -> cells of each level's box at the last call: grid_pos_many's estimates


class GridTooSparse(RuntimeError):
    """The bounding box of the candidate cells has more than GRID_MAX_CELLS cells (a few particles very far apart):
    the caller uses the sort-based device formulation instead."""


def grid_pos_many(pos, voxel_sizes, centralize=False, pad=0, hyst=0.1, center=None):
    """``grid_pos`` (utils/tools/losses.py:136-181) for SEVERAL voxel sizes over the same positions -- the coarse levels of one step
    (losses.py:266-272) -- via dmcf_grid_pos_bounds/_count/_write with TWO host round trips for all of them (cells of every level,
    then points of every level) instead of two per level.  -> [(points, (minp, dims) | None), ...].  A level whose box is too sparse
    for the dense cell table (GRID_MAX_CELLS_PER_POINT) takes the hashed table; GridTooSparse is no longer raised (round 6)."""
    import numpy as np
    L = _lib.lib()
    pos = _dev_f32(pos, "pos", 3)
    n = pos.shape[0]
    cen = _dev_f32(center.reshape(3), "center") if center is not None else None
    cflag = 1 if centralize else 0
    ws_bytes = L.dmcf_grid_pos_workspace_bytes(n)
    levels = []
    for v in voxel_sizes:
        vs = (ctypes.c_float * 3)(*[float(x) for x in np.asarray(v, dtype=np.float32).reshape(3)])
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=pos.device)
        _lib.check(L.dmcf_grid_pos_bounds(_ptr(pos), n, vs, cflag, _ptr(cen) if cen is not None else None, int(pad),
                                          float(hyst), _ptr(ws), ws_bytes, _stream()), "dmcf_grid_pos_bounds")
        levels.append(dict(vs=vs, ws=ws))
    def read_headers():
        hdrs = torch.stack([lv["ws"][0:64] for lv in levels]).cpu()
        for lv, hdr in zip(levels, hdrs):
            lv["center_host"] = hdr[40:52].view(torch.float32).tolist()
            lv["minp"], lv["dims"] = hdr[0:12].view(torch.int32).tolist(), hdr[12:24].view(torch.int32).tolist()
            lv["cells"] = int(hdr[24:32].view(torch.int64).item())
            lv["total"] = int(hdr[32:40].view(torch.int64).item())
            if lv["cells"] < 0:
                raise _lib.DmcfError("grid_pos: positions are not finite")
            # too sparse for the dense cell table (a few particles far from the rest): the hashed table (include/dmcf_hip.h)
            lv["sparse"] = lv["cells"] > min(GRID_MAX_CELLS, max(GRID_MIN_CELLS, GRID_MAX_CELLS_PER_POINT * n))

    noff = 1
    for x in np.asarray(voxel_sizes[0], dtype=np.float32).reshape(3):
        noff *= (2 + 2 * int(pad)) if x >= 1e-5 else 1
    hash_slots = 1 << max(int(4 * n * noff - 1).bit_length(), 10)  # >= twice the 2 n noff candidates

    def count(lv, capacity):
        if capacity < 0:  # hashed: 4 + 8 bytes per slot
            lv["table"] = torch.empty(3 * (-capacity), dtype=torch.int32, device=pos.device)
        else:
            lv["table"] = torch.empty(max(capacity, 1), dtype=torch.int32, device=pos.device)
        lv["capacity"] = capacity
        _lib.check(L.dmcf_grid_pos_count(_ptr(pos), n, lv["vs"], cflag, int(pad), float(hyst), _ptr(lv["ws"]), ws_bytes,
                                         _ptr(lv["table"]), capacity, _stream()), "dmcf_grid_pos_count")

    # A rollout asks for the same levels step after step and their boxes move slowly: with the cell count of the last call as an
    # estimate (+ 1/4, in size classes) the count pass runs BEFORE anything is read, and ONE host round trip brings header and point
    # count of every level; a level whose box outgrew its table is counted again with the exact size (a second round trip, rare).
    key = (tuple(tuple(float(x) for x in lv["vs"]) for lv in levels), bool(centralize), int(pad), float(hyst))
    est = _GRID_CELLS.get(key)
    if est is not None and len(est) == len(levels):
        for lv, c in zip(levels, est):  # (c < 0: the level was sparse at the last call)
            count(lv, -hash_slots if c < 0 else _size_class(c + c // 4))
        read_headers()
        again = [lv for lv in levels if lv["capacity"] >= 0 and (lv["sparse"] or lv["cells"] > lv["capacity"])]
        for lv in again:
            count(lv, -hash_slots if lv["sparse"] else lv["cells"])
        if again:
            read_headers()
    else:
        read_headers()  # (host round trip 1 of 2: every level's header)
        for lv in levels:
            count(lv, -hash_slots if lv["sparse"] else lv["cells"])
        read_headers()  # (host round trip 2 of 2: every level's point count)
    while len(_GRID_CELLS) > 32:  # (virtual ranks are threads sharing this dict: evict without iterating a dict another thread resizes)
        try:
            _GRID_CELLS.pop(next(iter(_GRID_CELLS)), None)
        except (RuntimeError, StopIteration):
            break
    _GRID_CELLS[key] = [-1 if lv["sparse"] else lv["cells"] for lv in levels]
    totals = [lv["total"] for lv in levels]
    res = []
    for lv, total in zip(levels, totals):
        out = torch.empty((total, 3), dtype=torch.float32, device=pos.device)
        if total:
            _lib.check(L.dmcf_grid_pos_write(_ptr(pos), n, lv["vs"], cflag, int(pad), float(hyst), _ptr(lv["ws"]), ws_bytes,
                                             _ptr(lv["table"]), lv["capacity"], _ptr(out), total, _stream()), "dmcf_grid_pos_write")
            if centralize and center is None:
                # out = float(cell) * voxel + mean(pos): every lattice built from these positions shares the centre exactly
                # (dmcf_amd/lattice.py; the lattice form of ContinuousConv uses it)
                from . import lattice
                # (the entry keeps ``pos`` alive: its address + version identify the family for as long as the entry exists)
                lattice.register(out, lv["ws"][40:52].view(torch.float32).clone(), [float(v) for v in lv["vs"]],
                                 ("mean", pos.data_ptr(), pos.shape[0], pos._version), lv["minp"], lv["dims"], keep=pos,
                                 center_host=lv["center_host"])
        res.append((out, (lv["minp"], lv["dims"]) if total else None))
    return res


def grid_pos(pos, voxel_size, centralize=False, pad=0, hyst=0.1, center=None, return_box=False):
    """Lattice points of ``grid_pos`` (utils/tools/losses.py:136-181) via dmcf_grid_pos_bounds/_count/_write.
    ``voxel_size``: 3 host floats; ``center``: optional [3] GPU tensor (lattice origin instead of the mean).
    ``return_box``: -> (points, (minp, dims) | None): the integer box of cells of THIS call (None for an empty result)."""
    out, box = grid_pos_many(pos, [voxel_size], centralize, pad, hyst, center)[0]
    return (out, box) if return_box else out


class GhostSelection:
    """A started ghost selection (dmcf_ghost_count has run): ``totals`` int64 [W, B] on the device = entries box b contributes to
    the list of width w.  :meth:`write` produces the lists once their sizes are known on the host."""

    def __init__(self, pos, boxes, widths2, ws, totals):
        self.pos, self.boxes, self.widths2, self.ws, self.totals = pos, boxes, widths2, ws, totals

    def write(self, sizes):
        """``sizes``: entries of list(w) per width (host ints: the row sums of ``totals``, read by the caller -- or known from the
        ranks that counted the same points).  -> [int64 tensor of list(w) for every w]: box-major, ascending point index."""
        L = _lib.lib()
        W, B = self.totals.shape
        sizes = [int(v) for v in sizes]
        starts = [0]
        for v in sizes[:-1]:
            starts.append(starts[-1] + v)
        rows = torch.empty(max(sum(sizes), 1), dtype=torch.int64, device=self.pos.device)
        c64 = ctypes.c_int64 * W
        w2 = (ctypes.c_float * W)(*self.widths2)
        _lib.check(L.dmcf_ghost_write(_ptr(self.pos), self.pos.shape[0], _ptr(self.boxes), B, w2, W, _ptr(rows), c64(*starts), c64(*sizes),
                                      _ptr(self.ws), self.ws.numel(), _stream()), "dmcf_ghost_write")
        return [rows[starts[w]: starts[w] + sizes[w]] for w in range(W)]


def ghost_select(pos, boxes, widths2):
    """dmcf_ghost_count: for every point of ``pos`` [n, 3] and every box of ``boxes`` [B, 6] (lo xyz, hi xyz; device float32) the
    number of ``widths2`` (host floats, DESCENDING) its squared distance to the box does not exceed.  ``widths2`` = [-1.0] selects
    OWNERSHIP instead (lo <= x < hi per axis): one list, the stable order of the points by owning box.  Returns a
    :class:`GhostSelection`."""
    L = _lib.lib()
    pos = _dev_f32(pos, "pos", 3)
    boxes = _dev_f32(boxes.reshape(-1, 6), "boxes", 6)
    B, W = boxes.shape[0], len(widths2)
    widths2 = [float(np.float32(v)) for v in widths2]
    nbytes = L.dmcf_ghost_workspace_bytes(pos.shape[0], B, W)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=pos.device)
    totals = torch.empty((W, B), dtype=torch.int64, device=pos.device)
    w2 = (ctypes.c_float * W)(*widths2)
    _lib.check(L.dmcf_ghost_count(_ptr(pos), pos.shape[0], _ptr(boxes), B, w2, W, _ptr(totals), _ptr(ws), nbytes, _stream()),
               "dmcf_ghost_count")
    return GhostSelection(pos, boxes, widths2, ws, totals)


def window_sum(points, queries, radius, window=None, ignore_query_point=False, hash_table=None):
    """dmcf_frs_window_sum: out[q] = sum_{|p - q| <= R} window(|p - q|^2 / R^2) (``window``: a WINDOWS key; None counts
    the neighbours, 'explicit' sums the squared distances).  The fused form of ``compute_density``
    (utils/tools/losses.py:285-306): the candidate scan of the search with the sum inside, no pair list."""
    L = _lib.lib()
    points = _dev_f32(points, "points", 3)
    queries = _dev_f32(queries, "queries", 3)
    radius = float(radius)
    if window not in WINDOWS:
        raise NotImplementedError(f"window {window!r}")
    n, m = points.shape[0], queries.shape[0]
    if hash_table is None or hash_table.n_queries_capacity < m or hash_table.points.data_ptr() != points.data_ptr() \
            or hash_table.radius != radius:
        hash_table = build_spatial_hash_table(points, radius, n_queries=m)
    nbytes = L.dmcf_frs_workspace_bytes(n, hash_table.n_queries_capacity)
    out = torch.empty(m, dtype=torch.float32, device=points.device)
    _lib.check(L.dmcf_frs_window_sum(_ptr(queries), m, n, radius, frs_flags(ignore_query_point), WINDOWS[window],
                                     _ptr(hash_table.workspace), nbytes, _ptr(out), _stream()), "dmcf_frs_window_sum")
    return out


def farthest_point_sample(npoint, inp):
    """Mirror of ``utils/tools/sampling.py: farthest_point_sample(npoint, inp)``: ``inp`` [1, n, 3] -> int32 [1, npoint]
    (dmcf_farthest_point_sample; the batch dimension of this path is always 1, utils/tools/losses.py:278-279)."""
    L = _lib.lib()
    if inp.dim() != 3 or inp.shape[0] != 1 or inp.shape[2] != 3:
        raise ValueError("farthest_point_sample expects a [1, n, 3] tensor")
    pts = _dev_f32(inp[0], "inp", 3)
    n, m = pts.shape[0], int(npoint)
    if m > 0 and n == 0:
        raise ValueError("cannot sample from an empty point set")
    nbytes = L.dmcf_fps_workspace_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=pts.device)
    idx = torch.empty(m, dtype=torch.int32, device=pts.device)
    _lib.check(L.dmcf_farthest_point_sample(_ptr(pts), n, m, _ptr(ws), nbytes, _ptr(idx), _stream()),
               "dmcf_farthest_point_sample")
    return idx.unsqueeze(0)


def gather_point(inp, idx):
    """Mirror of ``gather_point(inp [1, n, c], idx [1, m]) -> [1, m, c]`` (utils/tools/sampling.py)."""
    L = _lib.lib()
    if inp.dim() != 3 or inp.shape[0] != 1 or idx.dim() != 2 or idx.shape[0] != 1:
        raise ValueError("gather_point expects inp [1, n, c] and idx [1, m]")
    x = _dev_f32(inp[0], "inp")
    if idx.dtype != torch.int32 or not idx.is_cuda:
        raise TypeError("idx must be an int32 GPU tensor")
    ii = idx[0].contiguous()
    out = torch.empty((ii.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
    _lib.check(L.dmcf_gather_point(_ptr(x), _ptr(ii), ii.shape[0], x.shape[1], _ptr(out), _stream()), "dmcf_gather_point")
    return out.unsqueeze(0)
