"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/dmcf_hip.h declares (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip_lib():
    from dmcf_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "dmcf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dmcf_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(hip_lib):
    from dmcf_amd import _lib
    declared = _declared_symbols()
    assert declared, "no prototypes found in include/dmcf_hip.h"
    assert sorted(_lib.SYMBOLS) == declared
    for name in declared:
        assert hasattr(hip_lib, name), f"{name} declared in dmcf_hip.h but not exported"


def test_version_and_error_strings(hip_lib):
    assert hip_lib.dmcf_version() >= 100
    assert hip_lib.dmcf_error_string(0) == b"ok"
    assert b"workspace" in hip_lib.dmcf_error_string(-2)


def test_workspace_queries_and_host_validation(hip_lib):
    # host-side argument validation runs without a device
    assert hip_lib.dmcf_frs_workspace_bytes(1000, 1000) > 1000 * 16
    assert hip_lib.dmcf_frs_workspace_bytes(-1, 0) == 0
    assert hip_lib.dmcf_frs_build(None, 10, 0.1, None, 0, None) == -1  # null workspace -> DMCF_EINVAL
    assert hip_lib.dmcf_reduce_subarrays_sum(None, None, -1, None, None) == -1
    assert hip_lib.dmcf_points_aabb_workspace_bytes() >= 6 * 4
    assert hip_lib.dmcf_dense_forward(None, 10, 8, None, 16, None, None, None, None) == -1  # null operands -> DMCF_EINVAL
    assert hip_lib.dmcf_points_aabb(None, 10, None, None, 0, None) == -1  # no output -> DMCF_EINVAL
    from dmcf_amd._lib import CconvArgs
    a = CconvArgs()
    assert hip_lib.dmcf_cconv_forward(ctypes.byref(a), None, 0, None) == -1  # zero filter dims


def test_struct_layout_matches_header():
    # field order of struct dmcf_cconv_args in the header == ctypes mirror
    from dmcf_amd._lib import CconvArgs
    text = open(os.path.join(ROOT, "include", "dmcf_hip.h")).read()
    body = text[text.index("typedef struct dmcf_cconv_args {"):text.index("} dmcf_cconv_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"\b([a-z_]+)(?:\[5\])?;", body)
    assert names == [f[0] for f in CconvArgs._fields_]


def test_product_does_not_import_oracle():
    # the oracle is test infrastructure; nothing under dmcf_amd/ may reference it
    for dirpath, _, files in os.walk(os.path.join(ROOT, "dmcf_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "libdmcf_oracle" not in src, f


def test_ops_refuse_cpu_tensors(hip_lib):
    import torch
    from dmcf_amd import ops, _lib
    with pytest.raises(_lib.DmcfError):
        ops.fixed_radius_search(torch.zeros(4, 3), torch.zeros(4, 3), 0.5)


def test_block_diagonal_tile_mask():
    """filter_tile_mask of include/dmcf_hip.h: bit 4 * (c / 4) + o / 16 for the channels / outputs of every non-zero block."""
    from dmcf_amd import ops
    # conv200_2 (4 -> 32) + conv300_2 (8 -> 32): quad 0 feeds column tiles 0, 1; quads 1, 2 feed tiles 2, 3
    assert ops.block_diagonal_tile_mask([(0, 4, 0, 32), (4, 12, 32, 64)]) == 0b1100_1100_0011
    # blocks that straddle a quad / a tile set both
    assert ops.block_diagonal_tile_mask([(0, 6, 0, 8), (6, 8, 8, 24)]) == 0b0011_0001
    assert ops.block_diagonal_tile_mask([(0, 40, 0, 16)]) == 0 and ops.block_diagonal_tile_mask([(0, 4, 0, 80)]) == 0


def test_generated_splat_sources_are_current(tmp_path, monkeypatch):
    """dmcf_amd/csrc/cconv_*_splat*.inc and cconv_pair_*.inc are generated (tools/gen_cls_splat.py, tools/gen_z3_splat.py, tools/gen_pair_splat.py) and committed: the
    committed text must be what the generators write."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "dmcf_amd", "csrc")
    names = ["cconv_cls_splat.inc", "cconv_cls_splat8.inc", "cconv_z3_splat.inc"]
    committed = {n: open(os.path.join(csrc, n)).read() for n in names}
    real_join = os.path.join
    # the generators write next to the kernels: send their output to a scratch directory instead
    monkeypatch.setattr(os.path, "join", lambda *a: real_join(str(tmp_path), a[-1]) if a[-1] in names else real_join(*a))
    for gen in ("gen_cls_splat", "gen_z3_splat"):
        spec = importlib.util.spec_from_file_location(gen, real_join(root, "tools", gen + ".py"))
        spec.loader.exec_module(importlib.util.module_from_spec(spec))
    for n in names:
        assert open(real_join(str(tmp_path), n)).read() == committed[n], n
    # tools/gen_pair_splat.py (cconv_pair.hip) takes its output directory as an argument
    spec = importlib.util.spec_from_file_location("gen_pair_splat", real_join(root, "tools", "gen_pair_splat.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    gen.write_product(str(tmp_path))
    for n in gen.PRODUCT_FILES:
        assert open(real_join(str(tmp_path), n)).read() == open(real_join(csrc, n)).read(), n
    spec = importlib.util.spec_from_file_location("gen_p16_splat", real_join(root, "tools", "gen_p16_splat.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    gen.write_product(str(tmp_path))
    for n in gen.PRODUCT_FILES:
        assert open(real_join(str(tmp_path), n)).read() == open(real_join(csrc, n)).read(), n
