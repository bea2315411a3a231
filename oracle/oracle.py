"""numpy/ctypes face of the CPU oracle (test infrastructure only; parity unpinned).

Every function cites the reference call site whose contract it restates
(paths relative to /root/reference).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdmcf_oracle.so")
_lib = None

MAPPINGS = {"ball_to_cube_radial": 0, "ball_to_cube_volume_preserving": 1, "identity": 2}
INTERPOLATIONS = {"linear": 0, "linear_border": 1, "nearest_neighbor": 2}


def build(force=False):
    """Compile oracle/dmcf_oracle.c with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "dmcf_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libdmcf_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        c = ctypes
        L.dmcf_ref_hash_table_size.restype = c.c_int64
        L.dmcf_ref_hash_table_size.argtypes = [c.c_int64, c.c_double]
        L.dmcf_ref_build_spatial_hash_table.restype = c.c_int
        L.dmcf_ref_build_spatial_hash_table.argtypes = [c.c_void_p, c.c_int64, c.c_float, c.c_int64,
                                                        c.c_void_p, c.c_void_p]
        L.dmcf_ref_fixed_radius_search.restype = c.c_int64
        L.dmcf_ref_fixed_radius_search.argtypes = [c.c_void_p, c.c_int64, c.c_void_p, c.c_int64, c.c_float,
                                                   c.c_int, c.c_int64, c.c_void_p, c.c_void_p,
                                                   c.c_void_p, c.c_void_p, c.c_void_p, c.c_int]
        L.dmcf_ref_bruteforce_search.restype = c.c_int64
        L.dmcf_ref_bruteforce_search.argtypes = [c.c_void_p, c.c_int64, c.c_void_p, c.c_int64, c.c_float,
                                                 c.c_int, c.c_void_p, c.c_void_p, c.c_void_p]
        L.dmcf_ref_reduce_subarrays_sum.restype = c.c_int
        L.dmcf_ref_reduce_subarrays_sum.argtypes = [c.c_void_p, c.c_void_p, c.c_int64, c.c_void_p]
        for name in ("dmcf_ref_continuous_conv", "dmcf_ref_continuous_conv_f64"):
            f = getattr(L, name)
            f.restype = c.c_int
            f.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int64, c.c_float, c.c_void_p, c.c_int64,
                          c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int, c.c_int,
                          c.c_int, c.c_int, c.c_void_p]
        L.dmcf_ref_filter_coordinates.restype = c.c_int
        L.dmcf_ref_filter_coordinates.argtypes = [c.c_void_p, c.c_int64, c.c_float, c.c_void_p, c.c_int,
                                                  c.c_int, c.c_void_p]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


# ---------------------------------------------------------------------------------------------
# neighbour search -- utils/convolutions.py:207-210 (ctor), :354-358 (call);
#                     utils/tools/losses.py:296-298 (tuple-unpacked result)
# ---------------------------------------------------------------------------------------------
# Which hash bins a query visits (dmcf_oracle.c, DMCF_REF_BINS_*): "own+corners" = the query's own voxel and the 8 corner
# voxels of q +- R (the default), "corners" = the 8 corner voxels only (round 3's reading),
# "all" = the 27 voxels around the query: the set of the distance test, not a reading of the library.
BINS = {"corners": 0, "own+corners": 1, "all": 2}
_default_bins = "own+corners"


class search_bins:
    """``with oracle.search_bins("corners"): ...`` -- every search of the block (the model restatement's too) walks that
    bin set."""

    def __init__(self, bins):
        if bins not in BINS:
            raise ValueError(f"bins must be one of {sorted(BINS)}")
        self.bins = bins

    def __enter__(self):
        global _default_bins
        self.saved, _default_bins = _default_bins, self.bins
        return self

    def __exit__(self, *exc):
        global _default_bins
        _default_bins = self.saved
        return False


def fixed_radius_search(points, queries, radius, ignore_query_point=False,
                        hash_table_size_factor=1 / 64, bruteforce=False, bins=None):
    """-> (neighbors_index int32 [P], neighbors_row_splits int64 [m+1], neighbors_distance f32 [P])

    Restates ml3d.layers.FixedRadiusSearch(metric='L2', ignore_query_point, return_distances=True)
    (points, queries, radius): build_spatial_hash_table + fixed_radius_search of Open3D 0.15.2.
    Rows are in the oracle's order (ascending hash bin, then ascending point id); compare rows as
    sorted sets.  distances are squared L2.  ``bins``: see BINS (None = the current default).
    """
    bin_set = BINS[_default_bins if bins is None else bins]
    L = lib()
    points = _f32(points).reshape(-1, 3)
    queries = _f32(queries).reshape(-1, 3)
    n, m = points.shape[0], queries.shape[0]
    radius = float(np.float32(radius))
    rs = np.zeros(m + 1, dtype=np.int64)
    if bruteforce:
        total = L.dmcf_ref_bruteforce_search(_ptr(points), n, _ptr(queries), m, radius,
                                             int(ignore_query_point), _ptr(rs), None, None)
        assert total >= 0
        idx = np.empty(total, dtype=np.int32)
        dist = np.empty(total, dtype=np.float32)
        L.dmcf_ref_bruteforce_search(_ptr(points), n, _ptr(queries), m, radius, int(ignore_query_point),
                                     _ptr(rs), _ptr(idx), _ptr(dist))
        return idx, rs, dist
    size = L.dmcf_ref_hash_table_size(n, float(hash_table_size_factor))
    splits = np.zeros(size + 1, dtype=np.uint32)
    table = np.zeros(max(n, 1), dtype=np.uint32)
    err = L.dmcf_ref_build_spatial_hash_table(_ptr(points), n, radius, size, _ptr(splits), _ptr(table))
    assert err == 0, err
    total = L.dmcf_ref_fixed_radius_search(_ptr(points), n, _ptr(queries), m, radius,
                                           int(ignore_query_point), size, _ptr(splits), _ptr(table),
                                           _ptr(rs), None, None, bin_set)
    assert total >= 0, total
    idx = np.empty(total, dtype=np.int32)
    dist = np.empty(total, dtype=np.float32)
    L.dmcf_ref_fixed_radius_search(_ptr(points), n, _ptr(queries), m, radius, int(ignore_query_point),
                                   size, _ptr(splits), _ptr(table), _ptr(rs), _ptr(idx), _ptr(dist), bin_set)
    return idx, rs, dist


def canonical_rows(idx, row_splits, *per_pair):
    """Sort every CSR row by neighbour index (rows are sets; the order inside a row is
    implementation-defined in the reference).  Returns (idx_sorted, *per_pair_sorted)."""
    idx = np.asarray(idx)
    rs = np.asarray(row_splits)
    row_of = np.repeat(np.arange(len(rs) - 1, dtype=np.int64), np.diff(rs))
    order = np.lexsort((idx, row_of))
    return (idx[order],) + tuple(np.asarray(p)[order] for p in per_pair)


def reduce_subarrays_sum(values, row_splits):
    """models/pbf_model.py:450-453  o3dml.ops.reduce_subarrays_sum(values, row_splits)."""
    values = _f32(values)
    rs = np.ascontiguousarray(row_splits, dtype=np.int64)
    out = np.zeros(len(rs) - 1, dtype=np.float32)
    lib().dmcf_ref_reduce_subarrays_sum(_ptr(values), _ptr(rs), len(rs) - 1, _ptr(out))
    return out


# ---------------------------------------------------------------------------------------------
# window functions -- utils/tools/losses.py:8-44 (input q = d^2 / R^2)
# ---------------------------------------------------------------------------------------------
def window(typ, q, fac=1.0):
    q = np.asarray(q, dtype=np.float32)
    one = np.float32(1)
    if typ == "poly6":  # losses.py:11-12
        return np.float32(fac) * np.clip((one - q) ** 3, 0, 1).astype(np.float32)
    if typ == "cubic":  # losses.py:15-20
        s = np.sqrt(q)
        r = np.where(q <= 1, np.where(s <= 0.5, 6 * (s ** 3 - q) + 1, 2 * (1 - s) ** 3), 0.0)
        return (np.float32(fac) * np.float32(4) / np.float32(3) * r).astype(np.float32)
    if typ == "linear":  # losses.py:23-25
        return (np.float32(fac) * (one - np.sqrt(q))).astype(np.float32)
    if typ == "peak":  # losses.py:28-30
        s = np.sqrt(q)
        return (np.float32(fac) * (one - np.float32(2) * s + q)).astype(np.float32)
    if typ == "cubic_grad":  # losses.py:33-39
        s = np.sqrt(q)
        r = np.where(q <= 1, np.where(s <= 0.5, 18 * q - 12 * s, -6 * (1 - s) ** 2), 0.0)
        return (np.float32(fac) * np.float32(4) / np.float32(3) * r).astype(np.float32)
    if typ is None:
        return None
    raise NotImplementedError(typ)


# ---------------------------------------------------------------------------------------------
# continuous convolution -- utils/convolutions.py:414-431
# ---------------------------------------------------------------------------------------------
def continuous_conv(filters, out_positions, extents, inp_positions, inp_features, neighbors_index,
                    neighbors_row_splits, neighbors_importance=None, inp_importance=None,
                    align_corners=True, coordinate_mapping="ball_to_cube_volume_preserving",
                    interpolation="linear", normalize=False, offset=None, f64=False):
    """Restates ml3d.ops.continuous_conv for a scalar extent.  filters [D,H,W,Cin,Cout]."""
    if offset is not None:
        assert not np.any(np.asarray(offset)), "non-zero offset is never used by DMCF (convolutions.py:200-201)"
    filters = _f32(filters)
    assert filters.ndim == 5
    dims = np.asarray(filters.shape, dtype=np.int32)
    out_positions = _f32(out_positions).reshape(-1, 3)
    inp_positions = _f32(inp_positions).reshape(-1, 3)
    inp_features = _f32(inp_features).reshape(inp_positions.shape[0], int(dims[3]))  # (an empty input set is legal: every
    # boundary particle cropped away, pbf_model.py:330-336 -- all rows are empty and the result is zero)
    extent = float(np.float32(np.asarray(extents).reshape(-1)[0]))
    idx = np.ascontiguousarray(neighbors_index, dtype=np.int32)
    rs = np.ascontiguousarray(neighbors_row_splits, dtype=np.int64)
    nimp = None
    if neighbors_importance is not None and np.size(neighbors_importance) > 0:
        nimp = _f32(neighbors_importance)
        assert nimp.shape[0] == idx.shape[0]
    pimp = None
    if inp_importance is not None and np.size(inp_importance) > 0:
        pimp = _f32(inp_importance)
    m = out_positions.shape[0]
    out = np.zeros((m, int(dims[4])), dtype=np.float32)
    fn = lib().dmcf_ref_continuous_conv_f64 if f64 else lib().dmcf_ref_continuous_conv
    err = fn(_ptr(filters), _ptr(dims), _ptr(out_positions), m, extent, _ptr(inp_positions),
             inp_positions.shape[0], _ptr(inp_features), _ptr(pimp), _ptr(idx), _ptr(rs), _ptr(nimp),
             int(align_corners), MAPPINGS[coordinate_mapping], INTERPOLATIONS[interpolation],
             int(normalize), _ptr(out))
    assert err == 0, err
    return out


def filter_coordinates(rel, extent, kernel_size_zyx, align_corners=True,
                       coordinate_mapping="ball_to_cube_volume_preserving"):
    """Filter-array coordinates (x, y, z) of relative positions; for analytic mapping tests."""
    rel = _f32(rel).reshape(-1, 3)
    ks = np.asarray(kernel_size_zyx, dtype=np.int32)
    out = np.zeros_like(rel)
    lib().dmcf_ref_filter_coordinates(_ptr(rel), rel.shape[0], float(extent), _ptr(ks), int(align_corners),
                                      MAPPINGS[coordinate_mapping], _ptr(out))
    return out


def mirror_kernel(kernel, sym_axis):
    """utils/convolutions.py:410-412: full = concat([-k[::-1, ::-1, ::-1], k], axis=sym_axis)."""
    kernel = np.asarray(kernel)
    return np.concatenate([-kernel[::-1, ::-1, ::-1], kernel], axis=sym_axis)


class ContinuousConvRef:
    """Restates ContinuousConv.call (utils/convolutions.py:277-470) on numpy arrays for the flag set
    DMCF uses (models/pbf_model.py:197-224): scalar extents, L2 metric, optional window,
    normalize, symmetric (ASCC, two-pass form exactly as :433-458), bias, no activation."""

    def __init__(self, kernel, bias=None, window_function=None, ignore_query_points=False,
                 symmetric=False, sym_axis=2, normalize=False, align_corners=True,
                 coordinate_mapping="ball_to_cube_volume_preserving", interpolation="linear", f64=False):
        self.kernel = np.asarray(kernel, dtype=np.float32)
        self.bias = None if bias is None else np.asarray(bias, dtype=np.float32)
        self.window_function = window_function
        self.ignore_query_points = ignore_query_points
        self.symmetric = symmetric
        self.sym_axis = sym_axis
        self.normalize = normalize
        self.align_corners = align_corners
        self.coordinate_mapping = coordinate_mapping
        self.interpolation = interpolation
        self.f64 = f64

    def __call__(self, inp_features, inp_positions, out_positions, extents, nns=None):
        radius = np.float32(0.5) * np.float32(extents)  # :353
        if nns is None:
            nns = fixed_radius_search(inp_positions, out_positions, radius, self.ignore_query_points)
        idx, rs, dist = nns
        self.nns = nns
        if self.window_function is None:
            importance = None
        else:
            q = dist / (radius * radius)  # :361-362
            importance = window(self.window_function, q)  # :378-379
        kernel = self.kernel
        if self.symmetric:
            kernel = mirror_kernel(kernel, self.sym_axis)  # :410-412
        kw = dict(out_positions=out_positions, extents=extents, inp_positions=inp_positions,
                  neighbors_index=idx, neighbors_row_splits=rs, neighbors_importance=importance,
                  align_corners=self.align_corners, coordinate_mapping=self.coordinate_mapping,
                  interpolation=self.interpolation, normalize=self.normalize, f64=self.f64)
        inp_features = np.asarray(inp_features, dtype=np.float32)
        out = continuous_conv(kernel, inp_features=inp_features, **kw)  # :431
        if self.symmetric:  # :433-458
            weights = kernel.reshape(kernel.shape[0], kernel.shape[1], kernel.shape[2], 1, -1)
            ones = np.ones_like(inp_features[..., :1])
            w_values = continuous_conv(weights, inp_features=ones, **kw)
            res = w_values.reshape(-1, kernel.shape[-2], kernel.shape[-1])
            out = out + np.einsum("nc,nco->no", inp_features, res).astype(np.float32)
        if self.bias is not None:
            out = out + self.bias  # :466-467
        return out.astype(np.float32)


# ---------------------------------------------------------------------------------------------
# multi-scale point sets -- utils/tools/losses.py:136-181 (grid_pos), :249-284 (get_dilated_pos)
# ---------------------------------------------------------------------------------------------
def grid_pos(pos, voxel_size, centralize=False, pad=0, hyst=0.1):
    """Restates losses.py:136-181 in numpy float32.  Output order = first occurrence (tf.unique)."""
    pos = _f32(pos).reshape(-1, 3)
    voxel_size = _f32(voxel_size).reshape(3)
    center = None
    if centralize:
        # tf.reduce_mean (losses.py:138) reduces with Eigen's vectorised tree reduction: float32 result within
        # an ulp or two of the exact mean.  numpy's float32 sum over axis 0 accumulates naively (error ~N*eps),
        # so form the mean in float64 and round once.
        center = pos.mean(axis=0, dtype=np.float64).astype(np.float32)
        pos = pos - center
    vs = np.maximum(voxel_size, np.float32(1e-5))
    h = np.where(voxel_size >= 1e-5, np.float32(hyst), np.float32(0)).astype(np.float32)
    scaled = (pos / vs).astype(np.float32)
    dpos = np.concatenate([np.floor(scaled - h).astype(np.int32), np.floor(scaled + h).astype(np.int32)], axis=0)
    ranges = [np.arange(-pad, 2 + pad) if voxel_size[a] >= 1e-5 else np.arange(0, 1) for a in range(3)]
    off = np.stack(np.meshgrid(*ranges, indexing="ij"), axis=-1).reshape(1, -1, 3).astype(np.int32)
    dpos = (dpos[:, None, :] + off).reshape(-1, 3)
    minp = dpos.min(axis=0)
    maxp = (dpos.max(axis=0) - minp + 1).astype(np.int64)  # int64: the index space of a sparse scene exceeds 2^31
    idx = ((dpos - minp).astype(np.int64) * np.array([1, maxp[0], maxp[0] * maxp[1]], dtype=np.int64)).sum(-1)
    uniq, first = np.unique(idx, return_index=True)
    idx = uniq[np.argsort(first, kind="stable")]  # tf.unique keeps first-occurrence order
    gpos = np.stack([idx % maxp[0], idx // maxp[0] % maxp[1], idx // (maxp[0] * maxp[1])], axis=-1) + minp
    if centralize:
        return (gpos.astype(np.float32) * voxel_size + center).astype(np.float32)
    return (gpos.astype(np.float32) * voxel_size + voxel_size / np.float32(2)).astype(np.float32)


def get_dilated_pos(pos, strides, voxel_size, centralize=False, pad=0, hyst=0.1):
    """losses.py:249-284, voxel_size path only (every shipped multi-scale config sets voxel_size)."""
    out = []
    for stride in strides:
        if stride == 1:
            out.append(_f32(pos))
        else:
            out.append(grid_pos(pos, _f32(voxel_size) * np.float32(stride), centralize, pad, hyst))
    return out


def compute_density(out_pos, in_pos=None, radius=0.005, win=None):
    """Restates losses.py:285-306: sum over the points within ``radius`` of win(|in - out|^2 / radius^2)
    (win = a window name, or None = identity as in the reference's warning branch)."""
    out_pos = _f32(out_pos)
    in_pos = out_pos if in_pos is None else _f32(in_pos)
    idx, rs, _ = fixed_radius_search(in_pos, out_pos, radius)
    seg = np.repeat(np.arange(out_pos.shape[0]), np.diff(rs))
    diff = in_pos[idx] - out_pos[seg]
    q = ((diff ** 2).sum(axis=-1) / np.float32(radius) ** 2).astype(np.float32)
    w = q if win is None else window(win, q)
    return np.bincount(seg, weights=w.astype(np.float64), minlength=out_pos.shape[0]).astype(np.float32)


def compute_pressure(dens, rest_dens=3.5, stiffness=20.0):
    """losses.py:367-377 for a given density."""
    d = np.asarray(dens, dtype=np.float64)
    return np.maximum(stiffness * ((d / rest_dens) ** 7 - 1), 0).astype(np.float32)


def density_loss(gt, pred, gt_in=None, pred_in=None, radius=0.005, eps=0.01, win=None, use_max=False):
    """losses.py:380-398."""
    pred_dens = compute_density(pred, pred_in, radius, win)
    gt_dens = compute_density(gt, gt_in, radius, win)
    rest = gt_dens.max()
    if use_max:
        return np.float32(abs(pred_dens.max() - rest) / rest)
    return np.float32(np.maximum(pred_dens - rest - eps, 0).mean())


def point_sampling(feats, inp_pos, out_pos, extent, win=None, normalize=True, f64=False):
    """Restates PointSampling.call (convolutions.py:888-1061): continuous_conv with a 1x1x1 identity filter."""
    feats = _f32(feats)
    c = feats.shape[-1]
    radius = np.float32(0.5) * np.float32(extent)
    idx, rs, d = fixed_radius_search(_f32(inp_pos), _f32(out_pos), radius)
    imp = None if win is None else window(win, d / (radius * radius))
    k = np.eye(c, dtype=np.float32).reshape(1, 1, 1, c, c)
    return continuous_conv(k, _f32(out_pos), extent, _f32(inp_pos), feats, idx, rs, imp, align_corners=False,
                           coordinate_mapping="ball_to_cube_radial", interpolation="linear", normalize=normalize, f64=f64)


def farthest_point_sample(npoint, pts):
    """Restates farthestpointsamplingKernel (utils/tools/sampling.cu:125-182) for one point set: sample 0 = point 0,
    float32 squared distances (dx*dx + dy*dy) + dz*dz, running minimum, arg-max with the kernel's tie order
    (512 threads striding the points, first strict maximum per thread, tree reduction that keeps the lower slot):
    smallest (index mod 512), then smallest index."""
    pts = _f32(pts).reshape(-1, 3)
    n = pts.shape[0]
    idx = np.zeros(int(npoint), dtype=np.int32)
    temp = np.full(n, np.float32(1e38), dtype=np.float32)
    k = np.arange(n)
    order = np.lexsort((k, k & 511))  # candidates in tie-preference order
    old = 0
    for j in range(1, int(npoint)):
        d = pts - pts[old]
        d2 = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + d[:, 2] * d[:, 2]).astype(np.float32)
        temp = np.minimum(temp, d2)
        best = temp.max()
        cand = order[temp[order] == best]
        old = int(cand[0])
        idx[j] = old
    return idx


def get_dilated_pos_fps(pos, strides):
    """losses.py:274-282 (voxel_size is None): level k = n // stride farthest-point samples of level k-1.
    Returns (dilated_pos, idx) with idx[0] = None."""
    pos = _f32(pos)
    out, idx = [], []
    for stride in strides:
        if stride == 1:
            out.append(pos)
            idx.append(None)
        else:
            cnt = max(pos.shape[0] // stride, 1)
            idx.append(farthest_point_sample(cnt, out[-1]))
            out.append(np.ascontiguousarray(out[-1][idx[-1]]))
    return out, idx
