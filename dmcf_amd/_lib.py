"""ctypes binding of libdmcf_hip.so (the C ABI declared in include/dmcf_hip.h).

The product path has NO fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libdmcf_hip.so")
_lib = None


class DmcfError(RuntimeError):
    pass


class CconvArgs(ctypes.Structure):
    """struct dmcf_cconv_args (include/dmcf_hip.h)."""
    _fields_ = [
        ("filters", ctypes.c_void_p),
        ("filter_dims", ctypes.c_int32 * 5),
        ("sym_axis", ctypes.c_int32),
        ("out_positions", ctypes.c_void_p),
        ("n_out", ctypes.c_int64),
        ("inp_positions", ctypes.c_void_p),
        ("n_inp", ctypes.c_int64),
        ("inp_features", ctypes.c_void_p),
        ("inp_importance", ctypes.c_void_p),
        ("neighbors_index", ctypes.c_void_p),
        ("neighbors_row_splits", ctypes.c_void_p),
        ("neighbors_value", ctypes.c_void_p),
        ("extent", ctypes.c_float),
        ("window_fac", ctypes.c_float),
        ("window", ctypes.c_int32),
        ("coordinate_mapping", ctypes.c_int32),
        ("interpolation", ctypes.c_int32),
        ("flags", ctypes.c_int32),
        ("bias", ctypes.c_void_p),
        ("out", ctypes.c_void_p),
        ("n_pairs", ctypes.c_int64),
        ("neighbors_row_count", ctypes.c_void_p),
        ("filter_tile_mask", ctypes.c_uint32),
        ("row_length_hint", ctypes.c_int32),
    ]


class LatticeConvArgs(ctypes.Structure):
    """struct dmcf_lattice_conv_args (include/dmcf_hip.h)."""
    _fields_ = [
        ("filters", ctypes.c_void_p),
        ("filter_dims", ctypes.c_int32 * 5),
        ("inp_volume", ctypes.c_void_p),
        ("inp_min", ctypes.c_int32 * 3),
        ("inp_dims", ctypes.c_int32 * 3),
        ("out_table", ctypes.c_void_p),
        ("out_min", ctypes.c_int32 * 3),
        ("out_dims", ctypes.c_int32 * 3),
        ("n_out", ctypes.c_int64),
        ("inp_step", ctypes.c_int32),
        ("out_stride", ctypes.c_int32),
        ("out_phase", ctypes.c_int32 * 3),
        ("base_min", ctypes.c_int32 * 3),
        ("base_dims", ctypes.c_int32 * 3),
        ("rel_shift", ctypes.c_float * 3),
        ("voxel", ctypes.c_float * 3),
        ("offsets", ctypes.c_void_p),
        ("n_offsets", ctypes.c_int64),
        ("reach", ctypes.c_int32 * 3),
        ("extent", ctypes.c_float),
        ("window_fac", ctypes.c_float),
        ("window", ctypes.c_int32),
        ("coordinate_mapping", ctypes.c_int32),
        ("interpolation", ctypes.c_int32),
        ("flags", ctypes.c_int32),
        ("bias", ctypes.c_void_p),
        ("out", ctypes.c_void_p),
    ]


class CconvScatterArgs(ctypes.Structure):
    """struct dmcf_cconv_scatter_args (include/dmcf_hip.h)."""
    _fields_ = [
        ("filters", ctypes.c_void_p),
        ("filter_dims", ctypes.c_int32 * 5),
        ("out_positions", ctypes.c_void_p),
        ("n_out", ctypes.c_int64),
        ("inp_positions", ctypes.c_void_p),
        ("n_inp", ctypes.c_int64),
        ("inp_features", ctypes.c_void_p),
        ("t_index", ctypes.c_void_p),
        ("t_row_begin", ctypes.c_void_p),
        ("t_row_count", ctypes.c_void_p),
        ("t_capacity", ctypes.c_int64),
        ("plan", ctypes.c_void_p),
        ("block_cells", ctypes.c_int32),
        ("reach", ctypes.c_int32),
        ("extent", ctypes.c_float),
        ("window_fac", ctypes.c_float),
        ("window", ctypes.c_int32),
        ("flags", ctypes.c_int32),
        ("bias", ctypes.c_void_p),
        ("out", ctypes.c_void_p),
        ("error_flag", ctypes.c_void_p),
    ]


# names every entry point include/dmcf_hip.h declares (tests/test_abi.py cross-checks against the header)
SYMBOLS = [
    "dmcf_version", "dmcf_error_string", "dmcf_last_hip_error",
    "dmcf_frs_workspace_bytes", "dmcf_frs_build", "dmcf_frs_count", "dmcf_frs_write", "dmcf_frs_search_padded", "dmcf_frs_window_sum",
    "dmcf_cconv_workspace_bytes", "dmcf_cconv_forward", "dmcf_cconv_kernel_name",
    "dmcf_cconv_scatter_plan_bytes", "dmcf_cconv_scatter_plan", "dmcf_cconv_scatter_workspace_bytes", "dmcf_cconv_scatter_forward",
    "dmcf_lattice_conv_workspace_bytes", "dmcf_lattice_conv_forward",
    "dmcf_lattice_conv_batch_workspace_bytes", "dmcf_lattice_conv_forward_batch",
    "dmcf_reduce_subarrays_sum", "dmcf_points_aabb_workspace_bytes", "dmcf_points_aabb", "dmcf_dense_forward",
    "dmcf_fps_workspace_bytes", "dmcf_farthest_point_sample", "dmcf_gather_point",
    "dmcf_grid_pos_workspace_bytes", "dmcf_grid_pos_bounds", "dmcf_grid_pos_count", "dmcf_grid_pos_write",
    "dmcf_ghost_workspace_bytes", "dmcf_ghost_count", "dmcf_ghost_write",
]


def build(force=False):
    """Compile dmcf_amd/csrc/*.hip for gfx950 into dmcf_amd/libdmcf_hip.so (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_PKG, "csrc")
    args = ["make", "-C", csrc, "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DmcfError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    c = ctypes
    L.dmcf_version.restype = c.c_int
    L.dmcf_error_string.restype = c.c_char_p
    L.dmcf_error_string.argtypes = [c.c_int]
    L.dmcf_last_hip_error.restype = c.c_int
    L.dmcf_frs_workspace_bytes.restype = c.c_size_t
    L.dmcf_frs_workspace_bytes.argtypes = [c.c_int64, c.c_int64]
    L.dmcf_frs_build.restype = c.c_int
    L.dmcf_frs_build.argtypes = [c.c_void_p, c.c_int64, c.c_float, c.c_void_p, c.c_size_t, c.c_void_p]
    L.dmcf_frs_count.restype = c.c_int
    L.dmcf_frs_count.argtypes = [c.c_void_p, c.c_int64, c.c_int64, c.c_float, c.c_int, c.c_void_p, c.c_size_t,
                                 c.c_void_p, c.c_void_p]
    L.dmcf_frs_write.restype = c.c_int
    L.dmcf_frs_write.argtypes = [c.c_void_p, c.c_int64, c.c_int64, c.c_float, c.c_int, c.c_void_p, c.c_size_t,
                                 c.c_void_p, c.c_void_p, c.c_void_p, c.c_int64, c.c_void_p]
    L.dmcf_frs_search_padded.restype = c.c_int
    L.dmcf_frs_search_padded.argtypes = [c.c_void_p, c.c_int64, c.c_int64, c.c_float, c.c_int, c.c_void_p, c.c_size_t, c.c_int64,
                                         c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]
    L.dmcf_frs_window_sum.restype = c.c_int
    L.dmcf_frs_window_sum.argtypes = [c.c_void_p, c.c_int64, c.c_int64, c.c_float, c.c_int, c.c_int, c.c_void_p, c.c_size_t,
                                      c.c_void_p, c.c_void_p]
    L.dmcf_cconv_workspace_bytes.restype = c.c_size_t
    L.dmcf_cconv_workspace_bytes.argtypes = [c.POINTER(CconvArgs)]
    L.dmcf_cconv_forward.restype = c.c_int
    L.dmcf_cconv_forward.argtypes = [c.POINTER(CconvArgs), c.c_void_p, c.c_size_t, c.c_void_p]
    L.dmcf_cconv_kernel_name.restype = c.c_int
    L.dmcf_cconv_kernel_name.argtypes = [c.POINTER(CconvArgs), c.c_char_p, c.c_size_t]
    L.dmcf_cconv_scatter_plan_bytes.restype = c.c_size_t
    L.dmcf_cconv_scatter_plan_bytes.argtypes = [c.c_int64]
    L.dmcf_cconv_scatter_plan.restype = c.c_int
    L.dmcf_cconv_scatter_plan.argtypes = [c.c_void_p, c.c_int64, c.c_void_p, c.c_int64, c.c_float, c.c_float, c.c_int32, c.c_void_p,
                                          c.c_size_t, c.c_void_p]
    L.dmcf_cconv_scatter_workspace_bytes.restype = c.c_size_t
    L.dmcf_cconv_scatter_workspace_bytes.argtypes = [c.POINTER(CconvScatterArgs)]
    L.dmcf_cconv_scatter_forward.restype = c.c_int
    L.dmcf_cconv_scatter_forward.argtypes = [c.POINTER(CconvScatterArgs), c.c_void_p, c.c_size_t, c.c_void_p]
    L.dmcf_lattice_conv_workspace_bytes.restype = c.c_size_t
    L.dmcf_lattice_conv_workspace_bytes.argtypes = [c.POINTER(LatticeConvArgs)]
    L.dmcf_lattice_conv_forward.restype = c.c_int
    L.dmcf_lattice_conv_forward.argtypes = [c.POINTER(LatticeConvArgs), c.c_void_p, c.c_size_t, c.c_void_p]
    L.dmcf_lattice_conv_batch_workspace_bytes.restype = c.c_size_t
    L.dmcf_lattice_conv_batch_workspace_bytes.argtypes = [c.POINTER(LatticeConvArgs), c.c_int32]
    L.dmcf_lattice_conv_forward_batch.restype = c.c_int
    L.dmcf_lattice_conv_forward_batch.argtypes = [c.POINTER(LatticeConvArgs), c.c_int32, c.c_void_p, c.c_size_t, c.c_void_p]
    L.dmcf_dense_forward.restype = c.c_int
    L.dmcf_dense_forward.argtypes = [c.c_void_p, c.c_int64, c.c_int32, c.c_void_p, c.c_int32, c.c_void_p, c.c_void_p, c.c_void_p,
                                     c.c_void_p]
    L.dmcf_ghost_workspace_bytes.restype = c.c_size_t
    L.dmcf_ghost_workspace_bytes.argtypes = [c.c_int64, c.c_int32, c.c_int32]
    L.dmcf_ghost_count.restype = c.c_int
    L.dmcf_ghost_count.argtypes = [c.c_void_p, c.c_int64, c.c_void_p, c.c_int32, c.c_void_p, c.c_int32, c.c_void_p, c.c_void_p,
                                   c.c_size_t, c.c_void_p]
    L.dmcf_ghost_write.restype = c.c_int
    L.dmcf_ghost_write.argtypes = [c.c_void_p, c.c_int64, c.c_void_p, c.c_int32, c.c_void_p, c.c_int32, c.c_void_p, c.c_void_p,
                                   c.c_void_p, c.c_void_p, c.c_size_t, c.c_void_p]
    L.dmcf_points_aabb_workspace_bytes.restype = c.c_size_t
    L.dmcf_points_aabb_workspace_bytes.argtypes = []
    L.dmcf_points_aabb.restype = c.c_int
    L.dmcf_points_aabb.argtypes = [c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p, c.c_size_t, c.c_void_p]
    L.dmcf_reduce_subarrays_sum.restype = c.c_int
    L.dmcf_reduce_subarrays_sum.argtypes = [c.c_void_p, c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p]
    L.dmcf_fps_workspace_bytes.restype = c.c_size_t
    L.dmcf_fps_workspace_bytes.argtypes = [c.c_int64]
    L.dmcf_farthest_point_sample.restype = c.c_int
    L.dmcf_farthest_point_sample.argtypes = [c.c_void_p, c.c_int64, c.c_int64, c.c_void_p, c.c_size_t, c.c_void_p, c.c_void_p]
    L.dmcf_gather_point.restype = c.c_int
    L.dmcf_gather_point.argtypes = [c.c_void_p, c.c_void_p, c.c_int64, c.c_int, c.c_void_p, c.c_void_p]
    L.dmcf_grid_pos_workspace_bytes.restype = c.c_size_t
    L.dmcf_grid_pos_workspace_bytes.argtypes = [c.c_int64]
    f3 = c.POINTER(c.c_float)
    L.dmcf_grid_pos_bounds.restype = c.c_int
    L.dmcf_grid_pos_bounds.argtypes = [c.c_void_p, c.c_int64, f3, c.c_int, c.c_void_p, c.c_int, c.c_float, c.c_void_p,
                                       c.c_size_t, c.c_void_p]
    L.dmcf_grid_pos_count.restype = c.c_int
    L.dmcf_grid_pos_count.argtypes = [c.c_void_p, c.c_int64, f3, c.c_int, c.c_int, c.c_float, c.c_void_p, c.c_size_t,
                                      c.c_void_p, c.c_int64, c.c_void_p]
    L.dmcf_grid_pos_write.restype = c.c_int
    L.dmcf_grid_pos_write.argtypes = [c.c_void_p, c.c_int64, f3, c.c_int, c.c_int, c.c_float, c.c_void_p, c.c_size_t,
                                      c.c_void_p, c.c_int64, c.c_void_p, c.c_int64, c.c_void_p]
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        L = lib()
        raise DmcfError(f"{what}: {L.dmcf_error_string(rc).decode()} (code {rc}, hip error {L.dmcf_last_hip_error()})")
