"""CPU tests of the oracle itself (no GPU): the restated Open3D algorithms against brute force,
analytic known answers and the invariants of SURVEY.md section 4.  The reference ships no golden
vectors (parity unpinned), so these self-checks are what anchors the oracle."""
import numpy as np
import pytest


def _cloud(n, seed, dim=3, scale=1.0):
    rng = np.random.default_rng(seed)
    p = rng.uniform(-scale, scale, size=(n, 3)).astype(np.float32)
    if dim == 2:
        p[:, 2] = 0
    return p


@pytest.mark.parametrize("n,m,radius,dim", [(500, 300, 0.25, 3), (800, 800, 0.1, 2), (64, 1, 3.0, 3), (1, 50, 0.5, 3)])
@pytest.mark.parametrize("ignore", [False, True])
def test_hash_search_equals_bruteforce(oracle, n, m, radius, dim, ignore):
    pts = _cloud(n, 1, dim)
    qs = pts[:m].copy() if ignore and m <= n else _cloud(m, 2, dim)
    i0, r0, d0 = oracle.fixed_radius_search(pts, qs, radius, ignore)
    i1, r1, d1 = oracle.fixed_radius_search(pts, qs, radius, ignore, bruteforce=True)
    assert r0.dtype == np.int64 and i0.dtype == np.int32 and d0.dtype == np.float32
    np.testing.assert_array_equal(r0, r1)
    a, da = oracle.canonical_rows(i0, r0, d0)
    b, db = oracle.canonical_rows(i1, r1, d1)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(da, db)  # squared L2, bit-exact (same un-fused arithmetic)


def test_search_inclusive_radius_and_ignore_by_coordinates(oracle):
    # points exactly at distance R are neighbours (<=); duplicates of the query are dropped by
    # coordinate equality, not by index (SURVEY.md section 4 invariant 1)
    pts = np.array([[0, 0, 0], [0.5, 0, 0], [0, 0.5, 0], [0, 0, 0], [0.5000001, 0, 0], [1, 1, 1]], np.float32)
    qs = np.array([[0, 0, 0]], np.float32)
    idx, rs, d = oracle.fixed_radius_search(pts, qs, 0.5, False)
    assert sorted(idx.tolist()) == [0, 1, 2, 3]
    idx, rs, d = oracle.fixed_radius_search(pts, qs, 0.5, True)
    assert sorted(idx.tolist()) == [1, 2]
    np.testing.assert_array_equal(np.sort(d), np.float32([0.25, 0.25]))


def test_search_empty_inputs(oracle):
    idx, rs, d = oracle.fixed_radius_search(np.zeros((0, 3), np.float32), _cloud(5, 0), 0.3)
    assert idx.size == 0 and rs.tolist() == [0] * 6
    idx, rs, d = oracle.fixed_radius_search(_cloud(5, 0), np.zeros((0, 3), np.float32), 0.3)
    assert idx.size == 0 and rs.tolist() == [0]


def test_windows_known_values(oracle):
    q = np.float32([0.0, 0.25, 1.0, 1.5])
    np.testing.assert_allclose(oracle.window("poly6", q), [1.0, 0.421875, 0.0, 0.0], atol=1e-7)
    np.testing.assert_allclose(oracle.window("peak", q)[:3], [1.0, 0.25, 0.0], atol=1e-7)
    np.testing.assert_allclose(oracle.window("linear", q)[:3], [1.0, 0.5, 0.0], atol=1e-7)
    np.testing.assert_allclose(oracle.window("cubic", q)[:3], [4 / 3, 4 / 3 * 0.25, 0.0], atol=1e-6)


def test_mapping_on_axis_is_identity(oracle):
    # a neighbour on a coordinate axis: the volume preserving map is the identity along that axis,
    # so coordinate = (t/2 + 0.5) * (size-1)  (SURVEY.md section 8c "analytic cases")
    ext = 2.0  # radius 1
    t = np.float32([-1, -0.5, 0, 0.3, 1])
    for axis in range(3):
        rel = np.zeros((5, 3), np.float32)
        rel[:, axis] = t
        c = oracle.filter_coordinates(rel, ext, [4, 4, 4])
        expect = (t / 2 + 0.5) * 3
        np.testing.assert_allclose(c[:, axis], expect, atol=1e-6)
        for other in range(3):
            if other != axis:
                np.testing.assert_allclose(c[:, other], 1.5, atol=1e-6)


def test_mapping_ball_fills_cube(oracle):
    # points on the unit sphere land on the cube surface (max |coord| = 0.5 -> 0 or size-1)
    rng = np.random.default_rng(0)
    v = rng.normal(size=(2000, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    c = oracle.filter_coordinates(v, 2.0, [4, 4, 4]) / 3.0 - 0.5
    np.testing.assert_allclose(np.abs(c).max(axis=1), 0.5, atol=2e-6)
    # interior points stay inside, and the map is odd: L(-r) = -L(r)
    r = (v * rng.uniform(0, 1, size=(2000, 1))).astype(np.float32)
    c1 = oracle.filter_coordinates(r, 2.0, [4, 4, 4]) / 3.0 - 0.5
    c2 = oracle.filter_coordinates(-r, 2.0, [4, 4, 4]) / 3.0 - 0.5
    assert np.abs(c1).max() <= 0.5 + 1e-6
    np.testing.assert_allclose(c1, -c2, atol=1e-6)


def test_mapping_volume_preserving(oracle):
    # uniform samples in the ball map to (statistically) uniform samples in the cube
    rng = np.random.default_rng(1)
    p = rng.uniform(-1, 1, size=(400000, 3)).astype(np.float32)
    p = p[(p ** 2).sum(1) <= 1]
    c = oracle.filter_coordinates(p, 2.0, [2, 2, 2])  # size-1 = 1 -> coords in [0,1]
    hist, _ = np.histogramdd(c, bins=(4, 4, 4), range=[(0, 1)] * 3)
    frac = hist / hist.sum()
    np.testing.assert_allclose(frac, 1 / 64, rtol=0.05)


def _conv_case(oracle, seed=0, n=300, m=200, cin=5, cout=7, ks=(4, 4, 4), radius=0.35, dim=3):
    rng = np.random.default_rng(seed)
    inp = _cloud(n, seed, dim)
    out = _cloud(m, seed + 100, dim)
    feat = rng.normal(size=(n, cin)).astype(np.float32)
    filt = rng.uniform(-1, 1, size=(*ks, cin, cout)).astype(np.float32)
    idx, rs, d = oracle.fixed_radius_search(inp, out, radius)
    imp = oracle.window("poly6", d / np.float32(radius * radius))
    return inp, out, feat, filt, idx, rs, imp, np.float32(2 * radius)


def test_cconv_linearity_and_empty_rows(oracle):
    inp, out, feat, filt, _, _, _, ext = _conv_case(oracle)
    out = np.concatenate([out, np.float32([[5, 5, 5], [-7, 0, 0]])])  # two outputs with no neighbours
    idx, rs, d = oracle.fixed_radius_search(inp, out, ext / 2)
    imp = oracle.window("poly6", d / np.float32(ext / 2) ** 2)
    f = lambda F, W: oracle.continuous_conv(W, out, ext, inp, F, idx, rs, imp)
    y = f(feat, filt)
    np.testing.assert_allclose(f(2 * feat, filt), 2 * y, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(f(feat, 3 * filt), 3 * y, rtol=1e-5, atol=1e-5)
    empty = np.diff(rs) == 0
    assert empty.any()
    assert not y[empty].any()  # normalize=False: rows without neighbours are exactly zero


def test_cconv_constant_filter_is_windowed_sum(oracle):
    # SURVEY section 4 invariant 3: constant filter => out = (sum_j a_ij f_j) @ C, independent of the mapping
    inp, out, feat, filt, idx, rs, imp, ext = _conv_case(oracle, seed=3)
    C = np.random.default_rng(5).normal(size=(feat.shape[1], 7)).astype(np.float32)
    filt = np.broadcast_to(C, filt.shape).copy()
    y = oracle.continuous_conv(filt, out, ext, inp, feat, idx, rs, imp)
    row = np.repeat(np.arange(len(rs) - 1), np.diff(rs))
    s = np.zeros((len(rs) - 1, feat.shape[1]), np.float64)
    np.add.at(s, row, imp[:, None].astype(np.float64) * feat[idx])
    np.testing.assert_allclose(y, s @ C, rtol=2e-5, atol=2e-5)


def test_cconv_single_neighbour_known_answer(oracle):
    # one neighbour on the +x axis at t*R: out = a * f * lerp(filter row) computed by hand
    ks = (1, 1, 4)
    filt = np.arange(4 * 2 * 3, dtype=np.float32).reshape(1, 1, 4, 2, 3)
    inp = np.float32([[0.3, 0, 0]])
    out = np.float32([[0, 0, 0]])
    feat = np.float32([[2.0, -1.0]])
    idx, rs = np.int32([0]), np.int64([0, 1])
    imp = np.float32([0.5])
    y = oracle.continuous_conv(filt, out, 2.0, inp, feat, idx, rs, imp)
    x = (0.3 / 2 + 0.5) * 3  # 1.95 -> cells 1 and 2, weights 0.05 / 0.95
    g = 0.05 * filt[0, 0, 1] + 0.95 * filt[0, 0, 2]
    np.testing.assert_allclose(y[0], 0.5 * (feat[0] @ g), rtol=1e-5)


def test_cconv_f32_close_to_f64(oracle):
    inp, out, feat, filt, idx, rs, imp, ext = _conv_case(oracle, seed=7, cin=16, cout=8)
    y32 = oracle.continuous_conv(filt, out, ext, inp, feat, idx, rs, imp)
    y64 = oracle.continuous_conv(filt, out, ext, inp, feat, idx, rs, imp, f64=True)
    scale = np.abs(y64).max()
    assert np.abs(y32 - y64).max() <= 2e-5 * scale


def test_cconv_2d_degeneracy(oracle):
    # kernel [1,H,W] with z = 0: result independent of anything along z; filter z-size 1
    inp, out, feat, filt, idx, rs, imp, ext = _conv_case(oracle, seed=9, ks=(1, 8, 8), dim=2, radius=0.2, n=600, m=400)
    y = oracle.continuous_conv(filt, out, ext, inp, feat, idx, rs, imp)
    assert np.isfinite(y).all() and np.abs(y).max() > 0
    # embedding the same 2-D filter as the z-constant 3-D filter [2,H,W] gives the same answer
    filt3 = np.concatenate([filt, filt], axis=0)
    y3 = oracle.continuous_conv(filt3, out, ext, inp, feat, idx, rs, imp)
    np.testing.assert_allclose(y, y3, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("ks,sym_axis,dim", [((6, 6, 6), 1, 3), ((1, 8, 8), 1, 2), ((4, 4, 4), 2, 3), ((4, 4, 4), 0, 3)])
def test_ascc_momentum_and_fusion_identity(oracle, ks, sym_axis, dim):
    # SURVEY section 4 invariants 4 + 5
    rng = np.random.default_rng(11)
    n, cin, cout, radius = 400, 6, 3, 0.3
    pos = _cloud(n, 21, dim)
    feat = np.maximum(rng.normal(size=(n, cin)), 0).astype(np.float32)
    half = list(ks)
    half[sym_axis] //= 2
    k = rng.uniform(-1, 1, size=(*half, cin, cout)).astype(np.float32)
    conv = oracle.ContinuousConvRef(k, window_function="peak", ignore_query_points=True, symmetric=True,
                                    sym_axis=sym_axis)
    y = conv(feat, pos, pos, 2 * radius)
    idx, rs, d = conv.nns
    assert (np.diff(rs) > 0).mean() > 0.9
    terms = np.abs(y).sum(axis=0)
    assert np.all(np.abs(y.sum(axis=0)) <= 2e-5 * terms + 1e-6)  # sum_i out_i = 0
    # fused form: one CConv over pair features (f_j + f_i) with the mirrored kernel
    full = oracle.mirror_kernel(k, sym_axis)
    imp = oracle.window("peak", d / np.float32(radius * radius))
    row = np.repeat(np.arange(n), np.diff(rs))
    # emulate pair features by building a per-pair point set
    pair_pos = pos[idx]
    pair_feat = feat[idx] + feat[row]
    pair_idx = np.arange(len(idx), dtype=np.int32)
    y2 = oracle.continuous_conv(full, pos, 2 * radius, pair_pos, pair_feat, pair_idx, rs, imp)
    np.testing.assert_allclose(y, y2, rtol=1e-4, atol=1e-5 * np.abs(y).max())


def test_reduce_subarrays_sum(oracle):
    v = np.arange(10, dtype=np.float32)
    rs = np.int64([0, 3, 3, 10])
    np.testing.assert_array_equal(oracle.reduce_subarrays_sum(v, rs), [3, 0, 42])


def test_grid_pos_properties(oracle):
    rng = np.random.default_rng(0)
    pos = rng.uniform(0, 1, size=(500, 3)).astype(np.float32)
    vs = np.float32([0.1, 0.1, 0.1])
    g = oracle.grid_pos(pos, vs, centralize=True)
    # unique lattice points, every particle has its 8 surrounding corners present
    center = pos.mean(axis=0, dtype=np.float32)
    lat = np.rint((g - center) / vs).astype(np.int64)
    assert len(np.unique(lat, axis=0)) == len(lat)
    s = set(map(tuple, lat))
    cell = np.floor((pos - center) / vs).astype(np.int64)
    inner = np.abs((pos - center) / vs - np.rint((pos - center) / vs)).min(axis=1) > 0.11
    for c in cell[inner][:100]:
        for o in np.ndindex(2, 2, 2):
            assert tuple(c + np.array(o)) in s
    # 2-D: collapsed z axis
    pos2 = pos.copy()
    pos2[:, 2] = 0
    g2 = oracle.grid_pos(pos2, np.float32([0.1, 0.1, 0.0]), centralize=True)
    assert np.all(g2[:, 2] == 0)
    assert len(g2) < len(g)


def _edge_search(oracle, g, case, bins):
    pre = f"edge_{case}_"
    return oracle.fixed_radius_search(g[pre + "points"], g[pre + "queries"], float(g[pre + "radius"]), bool(g[pre + "ignore"]),
                                      hash_table_size_factor=float(g[pre + "factor"]), bins=bins)


def _edge_conv(oracle, g, case):
    pre = f"edge_{case}_"
    rel, filt = g[pre + "rel"], g[pre + "filt"]
    n = len(rel)
    return oracle.continuous_conv(filt, np.zeros((n, 3), np.float32), float(g[pre + "extent"]), rel, np.ones((n, 1), np.float32),
                                  np.arange(n, dtype=np.int32), np.arange(n + 1, dtype=np.int64), np.ones(n, np.float32))


def test_disputed_inputs_discriminate(oracle, tmp_path):
    """tools/capture_golden.py's disputed inputs (the ones an off-box run on TF 2.5 + Open3D 0.15.2 turns into golden
    vectors), run through the oracle HERE: every case separates the readings it is meant to separate, so one capture settles
    them; and the committed manifest is the script's."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tools", "capture_golden.py")
    man = json.loads(subprocess.run([sys.executable, script, "--manifest"], capture_output=True, text=True, check=True).stdout)
    with open(os.path.join(root, "tests", "golden", "open3d_golden.manifest.json")) as f:
        assert json.load(f) == man, "regenerate tests/golden/open3d_golden.manifest.json (tools/capture_golden.py --manifest)"
    path = str(tmp_path / "inputs.npz")
    subprocess.run([sys.executable, script, "--inputs-only", path], check=True)
    g = np.load(path)
    cases = sorted({k[len("edge_"):].rsplit("_", 1)[0] for k in g.files if k.endswith("_radius") or k.endswith("_rel")})
    assert all(f"edge_{c}_*" in man["keys"] for c in cases) and len(cases) >= 14

    def rows(case, bins=None, brute=False):
        pre = f"edge_{case}_"
        if brute:
            return np.diff(oracle.fixed_radius_search(g[pre + "points"], g[pre + "queries"], float(g[pre + "radius"]),
                                                      bool(g[pre + "ignore"]), bruteforce=True)[1])
        return np.diff(_edge_search(oracle, g, case, bins)[1])

    # (a) voxel midpoints: the two readings of the walk differ from each other and from the sphere, on the same queries
    for case in ("frs_midpoint", "frs_midpoint_ignore"):
        full, own, corners = rows(case, brute=True), rows(case, "own+corners"), rows(case, "corners")
        assert np.array_equal(rows(case, "all"), full)
        hit, hit_c = own < full, corners < full
        assert hit.sum() >= 5 and not (hit & ~hit_c).any() and (corners <= own).all() and (corners[hit_c] < own[hit_c]).any()
    assert rows("frs_midpoint", "own+corners")[120] < rows("frs_midpoint", brute=True)[120]  # z = 1.3, R = 0.1 itself
    # (b) the radius' edge far from the origin: both walks drop pairs of the sphere, the same ones
    full, own, corners = rows("frs_radius_edge_far", brute=True), rows("frs_radius_edge_far", "own+corners"), rows("frs_radius_edge_far", "corners")
    assert 0 < full.sum() - own.sum() < 0.01 * full.sum() and np.all(own <= full)
    # (c) one bin: nothing can hide; two bins and more: the midpoint rows shrink again
    for n in (10, 63, 64, 65, 127):
        assert oracle.lib().dmcf_ref_hash_table_size(n, 1 / 64) == 1
        for bins in ("own+corners", "corners"):
            assert np.array_equal(rows(f"frs_table_n{n}", bins), rows(f"frs_table_n{n}", brute=True)), (n, bins)
    assert oracle.lib().dmcf_ref_hash_table_size(128, 1 / 64) == 2 and oracle.lib().dmcf_ref_hash_table_size(200, 1 / 64) == 3
    # (with 2 - 3 bins the 8 corner voxels still reach every bin: these cases pin the clamp and the modulo, not the walk)
    assert rows("frs_table_n200", brute=True).sum() > 300
    # (d) the cap
    assert oracle.lib().dmcf_ref_hash_table_size(5000, 1.0e4) == 32 * 2 ** 20
    assert (rows("frs_table_cap", "own+corners") < rows("frs_table_cap", brute=True)).any()
    # (e) the map: both branches of both case distinctions are taken, the early-out and the border are reached
    rel = g["edge_map_444_rel"].astype(np.float64)
    cone = 1.25 * rel[:, 2] ** 2 - (rel[:, 0] ** 2 + rel[:, 1] ** 2)
    assert (cone > 0).sum() > 50 and (cone < 0).sum() > 50 and (np.abs(cone) < 1e-7).sum() > 20
    wedge = np.abs(rel[:, 1]) - np.abs(rel[:, 0])
    assert (wedge > 0).sum() > 50 and (wedge < 0).sum() > 50
    sq = (rel ** 2).sum(1)
    assert ((sq < 1e-12) & (sq > 0)).sum() >= 4 and (sq == 0).sum() >= 1 and ((sq > 1e-12) & (sq < 1e-9)).sum() >= 4
    coords = oracle.filter_coordinates(g["edge_map_444_rel"], float(g["edge_map_444_extent"]), (4, 4, 4))
    assert coords.min() < 0.0 and coords.max() > 3.0  # (unclamped coordinates:) the clamp at the filter's border acts on both sides
    for case in ("map_444", "map_188", "map_666"):
        y = _edge_conv(oracle, g, case)
        assert np.isfinite(y).all() and np.abs(y).max() > 0.1


def test_against_open3d_golden(oracle):
    """Pins the oracle to the real library once tools/capture_golden.py has been run off-box.  EVERY key of the capture is
    consumed (a key nobody looked at fails the test), every mismatch fails loudly; for the disputed inputs the failure
    message says which reading of the library the capture supports."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "open3d_golden.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/open3d_golden.npz not captured yet (parity unpinned, see DESIGN.md section 2)")
    _check_golden(oracle, path)


def _self_made_capture(oracle, path, bins="own+corners", with_cuda=True):
    """A file with the keys tools/capture_golden.py writes (without --reference), the ORACLE standing in for the library."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inputs = path + ".inputs.npz"
    subprocess.run([sys.executable, os.path.join(root, "tools", "capture_golden.py"), "--inputs-only", inputs], check=True)
    out = dict(np.load(inputs))
    g = dict(out)
    for case in sorted({k[len("edge_"):].rsplit("_", 1)[0] for k in g if k.endswith("_radius") or k.endswith("_rel")}):
        for tag in ("cpu", "cuda") if with_cuda else ("cpu",):
            if case.startswith("map_"):
                out[f"edge_{case}_out_{tag}"] = _edge_conv(oracle, g, case)
            else:
                idx, rs, d = _edge_search(oracle, g, case, bins if tag == "cpu" else "corners")
                out[f"edge_{case}_index_{tag}"], out[f"edge_{case}_row_splits_{tag}"], out[f"edge_{case}_distance_{tag}"] = idx, rs, d
    rng = np.random.default_rng(0)
    for name, dim, ks, radius, n, m in (("3d", 3, (4, 4, 4), 0.3, 800, 500), ("2d", 2, (1, 8, 8), 0.12, 900, 600), ("1d", 1, (1, 8, 1), 0.2, 300, 300)):
        pts = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
        qs = rng.uniform(-1, 1, size=(m, 3)).astype(np.float32)
        if dim <= 2:
            pts[:, 2] = 0
            qs[:, 2] = 0
        if dim == 1:
            pts[:, 0] = 0
            qs[:, 0] = 0
        for ign in (0, 1):
            q = pts[:m] if ign else qs
            idx, rs, d = oracle.fixed_radius_search(pts, q, radius, bool(ign))
            tag = f"{name}_ign{ign}"
            out[f"frs_{tag}_points"], out[f"frs_{tag}_queries"], out[f"frs_{tag}_radius"] = pts, q, np.float32(radius)
            out[f"frs_{tag}_index"], out[f"frs_{tag}_row_splits"], out[f"frs_{tag}_distance"] = idx, rs, d
        feat = rng.normal(size=(n, 6)).astype(np.float32)
        filt = rng.uniform(-1, 1, size=(*ks, 6, 5)).astype(np.float32)
        idx, rs, d = oracle.fixed_radius_search(pts, qs, radius)
        imp = oracle.window("poly6", d / np.float32(radius) ** 2)
        for mapping in ("ball_to_cube_volume_preserving", "ball_to_cube_radial", "identity"):
            out[f"cconv_{name}_{mapping}"] = oracle.continuous_conv(filt, qs, 2 * radius, pts, feat, idx, rs, imp, coordinate_mapping=mapping)
        out[f"cconv_{name}_feat"], out[f"cconv_{name}_filt"] = feat, filt
        out[f"cconv_{name}_index"], out[f"cconv_{name}_row_splits"], out[f"cconv_{name}_importance"] = idx, rs, imp
    v = rng.normal(size=100).astype(np.float32)
    rs = np.array([0, 10, 10, 55, 100], dtype=np.int64)
    out["rss_values"], out["rss_row_splits"], out["rss_out"] = v, rs, oracle.reduce_subarrays_sum(v, rs)
    np.savez(path, **out)
    return out


def test_golden_consumer_on_a_self_made_capture(oracle, tmp_path):
    """The consumer of the off-box capture, exercised NOW on a file with the capture's keys in which the oracle stands in for
    the library: it accepts its own reading, it reads every key, it FAILS (not skips) on a capture of the other reading and on
    a key it does not know -- so the day the real file arrives the test means something."""
    path = str(tmp_path / "capture.npz")
    out = _self_made_capture(oracle, path)
    _check_golden(oracle, path)
    np.savez(path, **out, edge_surprise_extra=np.zeros(3))
    with pytest.raises(AssertionError, match="no test consumed"):
        _check_golden(oracle, path)
    _self_made_capture(oracle, path, bins="corners", with_cuda=False)  # "the library walks the 8 corner voxels only"
    with pytest.raises(AssertionError, match="contradicts the oracle's reading"):
        _check_golden(oracle, path)
    out = _self_made_capture(oracle, path)
    out["edge_map_444_out_cpu"] = out["edge_map_444_out_cpu"].copy()
    out["edge_map_444_out_cpu"][7] *= np.float32(1.001)
    np.savez(path, **out)
    with pytest.raises(AssertionError, match="continuous_conv differs"):
        _check_golden(oracle, path)


def _check_golden(oracle, path):
    import os
    npz = np.load(path)
    seen = set()

    class Tracked:
        files = npz.files

        def __getitem__(self, k):
            seen.add(k)
            return npz[k]

        def __contains__(self, k):
            return k in npz.files
    g = Tracked()
    for name in ("3d", "2d", "1d"):
        for ign in (0, 1):
            tag = f"{name}_ign{ign}"
            idx, rs, d = oracle.fixed_radius_search(g[f"frs_{tag}_points"], g[f"frs_{tag}_queries"],
                                                    float(g[f"frs_{tag}_radius"]), bool(ign))
            np.testing.assert_array_equal(rs, g[f"frs_{tag}_row_splits"])
            a, da = oracle.canonical_rows(idx, rs, d)
            b, db = oracle.canonical_rows(g[f"frs_{tag}_index"], g[f"frs_{tag}_row_splits"], g[f"frs_{tag}_distance"])
            np.testing.assert_array_equal(a, b)
            np.testing.assert_allclose(da, db, rtol=1e-6)
        for mapping in ("ball_to_cube_volume_preserving", "ball_to_cube_radial", "identity"):
            y = oracle.continuous_conv(g[f"cconv_{name}_filt"], g[f"frs_{name}_ign0_queries"],
                                       2 * float(g[f"frs_{name}_ign0_radius"]), g[f"frs_{name}_ign0_points"],
                                       g[f"cconv_{name}_feat"], g[f"cconv_{name}_index"], g[f"cconv_{name}_row_splits"],
                                       g[f"cconv_{name}_importance"], coordinate_mapping=mapping)
            ref = g[f"cconv_{name}_{mapping}"]
            assert np.abs(y - ref).max() <= 1e-5 * np.abs(ref).max()
    np.testing.assert_allclose(oracle.reduce_subarrays_sum(g["rss_values"], g["rss_row_splits"]), g["rss_out"], rtol=1e-6)
    # ---- the disputed inputs
    cases = sorted({k[len("edge_"):].rsplit("_", 1)[0] for k in g.files
                    if k.startswith("edge_") and (k.endswith("_radius") or k.endswith("_rel"))})
    verdicts = []
    for case in cases:
        pre = f"edge_{case}_"
        for tag in ("cpu", "cuda"):
            if "rel" in case or case.startswith("map_"):
                if pre + "out_" + tag not in g:
                    continue
                y, ref = _edge_conv(oracle, g, case), g[pre + "out_" + tag]
                assert y.shape == ref.shape
                err = np.abs(y - ref).max() / np.abs(ref).max()
                assert err <= 1e-5, f"{case} [{tag}]: continuous_conv differs from the library by {err:.2e} (row {int(np.abs(y - ref).max(1).argmax())})"
                continue
            if pre + "row_splits_" + tag not in g:
                continue
            ref = (g[pre + "index_" + tag], g[pre + "row_splits_" + tag], g[pre + "distance_" + tag])
            match = {}
            for bins in ("own+corners", "corners", "all"):
                idx, rs, d = _edge_search(oracle, g, case, bins)
                ok = np.array_equal(rs, ref[1])
                if ok:
                    a, da = oracle.canonical_rows(idx, rs, d)
                    b, db = oracle.canonical_rows(*ref)
                    ok = np.array_equal(a, b) and np.array_equal(da, db)
                match[bins] = ok
            verdicts.append((case, tag, match))
    # the CPU path is the contract (BASELINE.json north_star): the oracle's DEFAULT reading must reproduce every CPU capture;
    # a CUDA capture must be reproduced by one of the readings (its 8-slot bin list is a different program)
    lines = [f"{c} [{t}]: " + ", ".join(f"{b}={'ok' if ok else 'DIFFERS'}" for b, ok in m.items()) for c, t, m in verdicts]
    bad = [line for (c, t, m), line in zip(verdicts, lines) if not (m["own+corners"] if t == "cpu" else any(m.values()))]
    assert not bad, "the capture contradicts the oracle's reading of open3d's search:\n" + "\n".join(lines)
    for k in g.files:  # inputs of the edge cases were read through _edge_search / _edge_conv
        if k.startswith("edge_") and k.rsplit("_", 1)[1] in ("points", "queries", "radius", "ignore", "factor", "rel", "extent", "filt"):
            seen.add(k)
    if "ascc_out" in g:  # captured with --reference: the reference's own ASCC layer, grid_pos and a 10-step rollout
        conv = oracle.ContinuousConvRef(g["ascc_kernel"], window_function="peak", ignore_query_points=True, symmetric=True,
                                        sym_axis=1)
        y = conv(g["ascc_feat"], g["ascc_pos"], g["ascc_pos"], 2 * float(g["ascc_radius"]))
        assert np.abs(y - g["ascc_out"]).max() <= 1e-5 * np.abs(g["ascc_out"]).max()
        for stride in (2, 4):
            for central in (0, 1):
                got = oracle.grid_pos(g["gridpos_cloud"], np.float32([0.025 * stride] * 3), centralize=bool(central))
                np.testing.assert_array_equal(got, g[f"gridpos_s{stride}_c{central}"])  # same points in tf.unique order
        from oracle.model_ref import ModelRef
        from tools import configs
        ref = ModelRef(configs.LIQUID3D, dict(np.load(os.path.join(os.path.dirname(path), "liquid3d_weights.npz"))))
        state = [g["rollout_pos"][0], g["rollout_vel0"], None, None, g["rollout_box"], g["rollout_box_normals"]]
        for t in range(1, g["rollout_pos"].shape[0]):
            pos, vel = ref.step(state)
            assert np.abs(pos - g["rollout_pos"][t]).max() <= 1e-5 * np.abs(g["rollout_pos"][t]).max(), f"rollout step {t}"
            state = [g["rollout_pos"][t], vel] + state[2:]  # per-step parity from the reference's own states
        seen.update(("rollout_vel_last", "rollout_seconds_per_step"))  # (informational: the reference's own wall time)
    left = sorted(set(g.files) - seen)
    assert not left, f"keys of the capture no test consumed: {left}"


def test_column_fixture_from_the_reference_generator(oracle):
    """tests/golden/column_test.npz holds the two test scenes of configs/column/hrnet.yml as the REFERENCE's own generator
    (datasets/column_gen.py, seed 44) produces them: their structure, and three oracle steps of the config's HRNet on them."""
    import os
    from oracle.model_ref import ModelRef
    from tools import configs, scenes
    fix = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "column_test.npz"))
    assert fix["s0_pos"].shape == (200, 1, 3) and fix["s1_pos"].shape == (200, 5, 3) and fix["s1_box"].shape == (2, 3)
    for s in (0, 1):
        pos, vel, grav = fix[f"s{s}_pos"], fix[f"s{s}_vel"], fix[f"s{s}_grav"]
        assert np.all(pos[..., 0] == 0) and np.all(pos[..., 2] == 0)          # a 1-D column along y
        assert np.allclose(grav, [0.0, -10.0, 0.0])                            # gravity / res * res
        assert pos[0, :, 1].min() > fix[f"s{s}_box"][:, 1].max()               # released above the two boundary points
        assert pos[50, :, 1].mean() < pos[0, :, 1].mean()                      # ... and falling
        cfg = dict(configs.COLUMN_HRNET)
        ref = ModelRef(cfg, scenes.random_weights(cfg, seed=2))
        state = [pos[0], vel[0], np.broadcast_to(grav[0], pos[0].shape).astype(np.float32).copy(), None,
                 fix[f"s{s}_box"], fix[f"s{s}_box_normals"]]
        for _ in range(3):
            p, v = ref.step(state)
            assert np.isfinite(p).all() and p.shape == pos[0].shape
            state = [p, v] + state[2:]
