#!/usr/bin/env python
"""Capture TRUE golden vectors from the reference's own operator library.

Run this ONCE on any machine with ``tensorflow==2.5`` and ``open3d==0.15.2`` (the reference's
requirements.txt); it does not need the reference repository.  It writes ``tests/golden/open3d_golden.npz``
with inputs and outputs of the three Open3D operators on the DMCF hot path for the flag sets DMCF uses.
Commit the file: ``tests/test_oracle.py::test_against_open3d_golden`` then pins the oracle (and through it
the HIP path) to the real library and the "parity unpinned" caveat in DESIGN.md can be dropped.

With ``--reference <path to a tum-pbs/DMCF checkout>`` it additionally captures, through the REFERENCE'S OWN Python
code (imported from that checkout, never copied):
  * the ASCC layer -- ``utils.convolutions.ContinuousConv(symmetric=True, sym_axis=1, window peak)``, i.e. the mirror of
    :410-412 and the second continuous_conv + matmul of :433-458 -- on a seeded cloud (``ascc_*``),
  * ``utils.tools.losses.grid_pos`` (:136-181) for two strides with and without ``centralize`` (``gridpos_*``),
  * a 10-step rollout of the Liquid3d SymNet with the shipped checkpoint on the canyon scene, driven as run_sample.py
    drives it (``rollout_*``: the positions after every step, and the wall time per step the reference logs).
``tests/test_oracle.py::test_against_open3d_golden`` and ``tests/test_gpu_model.py::test_against_reference_rollout_golden``
consume whatever the file holds.

The DISPUTED INPUTS (``edge_inputs``; listed by ``--manifest``, committed as tests/golden/open3d_golden.manifest.json) are
built so that ONE run settles every constant the oracle recalls from the library (oracle/dmcf_oracle.c, "EXT" tags): which
hash bins a query visits (own voxel + 8 corners, or the 8 corners alone), what happens a rounding step from a voxel's middle
and at the radius' edge far from the origin, the clamps of the table size, both branches of the sphere -> cylinder map, the
``sq_norm < 1e-12`` early-out and the clamp at the filter's border -- each on the CPU device and, when the machine has one, on
a CUDA device (keys ``..._cpu`` / ``..._cuda``).  ``--inputs-only`` writes the inputs without the libraries: the build container
runs the oracle on them (tests/test_oracle.py::test_disputed_inputs_discriminate) to show that the cases DO separate the
readings.

Nothing else in this script can run in the build container or on the GPU box (neither package exists for
ROCm 7 / Python 3.10); it is provided so the gap can be closed off-box.
"""
import argparse
import os
import sys
import time

import numpy as np


def _midpoint_queries(rng, radius, m, lo=0.2, hi=1.8):
    """m queries in [lo, hi]^3 of which the first ones sit at (k + 1/2) * 2R -+ {0, 1, 2} ulps on one axis (every such middle
    inside the range, on every axis): some of them hit the double rounding of floor(fl(q -+ R) / 2R) (e.g. z = 1.3 with
    R = 0.1: the corner voxels of that axis come out two apart)."""
    R = np.float32(radius)
    qs = rng.uniform(lo, hi, size=(m, 3)).astype(np.float32)
    mids = (np.arange(0, 64, dtype=np.float32) + np.float32(0.5)) * (np.float32(2) * R)
    mids = mids[(mids > lo) & (mids < hi)]
    k = 0
    for a in range(3):
        for mval in mids:
            for step in (-2, -1, 0, 1, 2):
                v = mval
                for _ in range(abs(step)):
                    v = np.nextafter(v, np.float32(np.inf if step > 0 else -np.inf), dtype=np.float32)
                qs[k, a] = v
                k += 1
    assert k <= m
    return qs


def edge_inputs():
    """The disputed inputs, as numpy arrays only (no TensorFlow / Open3D needed): {case: dict}.  Search cases hold ``points``,
    ``queries``, ``radius``, ``ignore``, ``factor`` (hash_table_size_factor of the call; the layer's default is 1/64); mapping
    cases hold ``rel`` (neighbour - query, one neighbour per query), ``extent``, ``filt``."""
    rng = np.random.default_rng(2024)
    cases = {}
    R = 0.1
    pts = rng.uniform(0.0, 2.0, size=(20000, 3)).astype(np.float32)
    # (a) a query a rounding step from the MIDDLE of a hash voxel (z = 1.3, R = 0.1 and its relatives on every axis)
    qs = _midpoint_queries(rng, R, 600)
    qs[120] = np.float32([0.77, 0.93, 1.3])  # the query the 1M-particle rollout found
    cases["frs_midpoint"] = dict(points=pts, queries=qs, radius=np.float32(R), ignore=False, factor=1 / 64)
    cases["frs_midpoint_ignore"] = dict(points=np.concatenate([qs, pts]), queries=qs, radius=np.float32(R), ignore=True, factor=1 / 64)
    # (b) pairs at distance R within rounding, 4000 away from the origin (one ulp of a coordinate is 0.5 % of R)
    a = rng.uniform(-0.5, 0.5, size=(3000, 3)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, size=(3000, 3)).astype(np.float32) + np.float32([4000.0, -2500.0, 900.0])
    far = np.concatenate([a, b])
    cases["frs_radius_edge_far"] = dict(points=far, queries=far[::3].copy(), radius=np.float32(0.05), ignore=False, factor=1 / 64)
    # (c) the table-size clamp from below: n * factor < 1 -> ONE bin, every query visits every point (no voxel can hide);
    #     n = 63 / 64 / 65 / 127 / 128 straddle the first two sizes.  The midpoint queries ride along: with one bin their rows
    #     must be complete, with two bins they may not be.
    for n in (10, 63, 64, 65, 127, 128, 200):
        p = rng.uniform(1.0, 1.6, size=(n, 3)).astype(np.float32)  # (a small cube: a handful of neighbours per row even at n = 200)
        q = _midpoint_queries(rng, R, 150, 1.05, 1.55)
        cases[f"frs_table_n{n}"] = dict(points=p, queries=q, radius=np.float32(R), ignore=False, factor=1 / 64)
    # (d) ... and from above: factor * n > 32 * 2^20 clamps to 2^25 bins (134 MB of uint32 row splits)
    p = rng.uniform(0.0, 2.0, size=(5000, 3)).astype(np.float32)
    cases["frs_table_cap"] = dict(points=p, queries=_midpoint_queries(rng, R, 300), radius=np.float32(R), ignore=False, factor=1.0e4)
    # (e) the ball -> cube map, one neighbour per query, feature 1, importance 1: out row = trilinear lookup of the filter.
    #     Both sides of the cone 5/4 z^2 = x^2 + y^2 (sphere -> cylinder), both sides of |y| = |x| (cylinder -> cube), the
    #     sq_norm < 1e-12 early-out (|rel| < 1e-6), the axes, and |rel| = extent / 2 exactly (clamp at the filter's border).
    extent = np.float32(0.5)
    rel = []
    for rho in (0.05, 0.2, 0.2499):
        for phi in np.linspace(0.0, 2.0 * np.pi, 16, endpoint=False) + 0.013:
            for sign in (1.0, -1.0):
                for eps in (-1e-3, -1e-6, 0.0, 1e-6, 1e-3):
                    # on the cone: z^2 = 4/5 (x^2 + y^2)  <=>  tan(theta) = sqrt(5) / 2
                    t = np.arctan(np.sqrt(5.0) / 2.0) + eps
                    rel.append([rho * np.sin(t) * np.cos(phi), rho * np.sin(t) * np.sin(phi), sign * rho * np.cos(t)])
    for rho in (0.1, 0.25):
        for z in (-0.3, 0.0, 0.6):
            for quad in range(4):
                for eps in (-1e-6, 0.0, 1e-6):
                    ang = np.pi / 4 + quad * np.pi / 2 + eps  # |y| = |x|
                    c = np.sqrt(max(1.0 - z * z, 0.0))
                    rel.append([rho * c * np.cos(ang), rho * c * np.sin(ang), rho * z])
    for mag in (0.0, 1e-8, 9e-7, 1.1e-6, 1e-5):  # around sqrt(1e-12)
        for axis in range(3):
            v = [0.0, 0.0, 0.0]
            v[axis] = mag
            rel.append(v)
        rel.append([mag / np.sqrt(3.0)] * 3)
    for axis in range(3):  # exactly on the ball's surface and a step beyond / inside, per axis and on the diagonal
        for mag in (np.float32(0.25), np.nextafter(np.float32(0.25), np.float32(1)), np.nextafter(np.float32(0.25), np.float32(0))):
            for sign in (1.0, -1.0):
                v = [0.0, 0.0, 0.0]
                v[axis] = sign * float(mag)
                rel.append(v)
    d = 0.25 / np.sqrt(3.0)
    rel += [[d, d, d], [-d, d, -d], [d * 1.0001, d * 1.0001, d * 1.0001]]
    rel = np.asarray(rel, np.float32)
    for name, ks in (("map_444", (4, 4, 4)), ("map_188", (1, 8, 8)), ("map_666", (6, 6, 6))):
        filt = rng.uniform(-1, 1, size=(*ks, 1, 8)).astype(np.float32)
        r = rel.copy()
        if ks[0] == 1:
            r[:, 2] = 0
        cases[name] = dict(rel=r, extent=extent, filt=filt)
    return cases


# what the capture holds and which test consumes it (tests/golden/open3d_golden.manifest.json is generated from this)
def manifest():
    m = {}
    for name, dim in (("3d", 3), ("2d", 2), ("1d", 1)):
        for ign in (0, 1):
            m[f"frs_{name}_ign{ign}_*"] = f"FixedRadiusSearch on a seeded {dim}-D cloud, ignore_query_point={bool(ign)}"
        m[f"cconv_{name}_*"] = "continuous_conv with DMCF's flags, three coordinate mappings"
    m["rss_*"] = "reduce_subarrays_sum"
    for case, c in edge_inputs().items():
        if "rel" in c:
            m[f"edge_{case}_*"] = ("continuous_conv on single-neighbour rows: both sides of the sphere->cylinder cone and of |y|=|x|, "
                                   "sq_norm < 1e-12, the ball's surface (clamp at the filter border); filter %s" % (c["filt"].shape[:3],))
        else:
            m[f"edge_{case}_*"] = ("FixedRadiusSearch, %d points, %d queries, R=%g, ignore=%s, hash_table_size_factor=%g; outputs per "
                                   "device (_cpu, _cuda)" % (len(c["points"]), len(c["queries"]), float(c["radius"]), c["ignore"], c["factor"]))
    m["ascc_* / gridpos_* / rollout_*"] = "--reference: the reference's ASCC layer, grid_pos and a 10-step Liquid3d rollout"
    return {"consumers": ["tests/test_oracle.py::test_against_open3d_golden", "tests/test_gpu_model.py::test_against_reference_rollout_golden",
                          "tests/test_gpu_ops.py::test_against_open3d_golden_hip"], "keys": m}


def capture_edges(out):
    """Run the disputed inputs through the library on every device it has."""
    import open3d.ml.tf as ml3d
    import tensorflow as tf
    devices = [("cpu", "/CPU:0")] + ([("cuda", "/GPU:0")] if tf.config.list_physical_devices("GPU") else [])
    for case, c in edge_inputs().items():
        for k, v in c.items():
            out[f"edge_{case}_{k}"] = np.asarray(v)
        for tag, dev in devices:
            with tf.device(dev):
                if "rel" in c:
                    rel, filt = c["rel"], c["filt"]
                    n = len(rel)
                    y = ml3d.ops.continuous_conv(filters=filt, out_positions=np.zeros((n, 3), np.float32),
                                                 extents=tf.constant([[float(c["extent"])]], tf.float32), offset=tf.zeros((3,)),
                                                 inp_positions=rel, inp_features=np.ones((n, 1), np.float32),
                                                 inp_importance=tf.ones((0,), tf.float32),
                                                 neighbors_index=np.arange(n, dtype=np.int32),
                                                 neighbors_row_splits=np.arange(n + 1, dtype=np.int64),
                                                 neighbors_importance=tf.ones((n,), tf.float32), align_corners=True,
                                                 coordinate_mapping="ball_to_cube_volume_preserving", interpolation="linear",
                                                 normalize=False)
                    out[f"edge_{case}_out_{tag}"] = y.numpy()
                else:
                    frs = ml3d.layers.FixedRadiusSearch(metric="L2", ignore_query_point=bool(c["ignore"]), return_distances=True)
                    res = frs(c["points"], c["queries"], float(c["radius"]), hash_table_size_factor=float(c["factor"]))
                    out[f"edge_{case}_index_{tag}"] = res.neighbors_index.numpy()
                    out[f"edge_{case}_row_splits_{tag}"] = res.neighbors_row_splits.numpy()
                    out[f"edge_{case}_distance_{tag}"] = res.neighbors_distance.numpy()


def capture_reference(ref_path, out, rng):
    """ASCC, grid_pos and a rollout through the reference's own modules (needs TF 2.5 + Open3D 0.15.2 + the checkout)."""
    import tensorflow as tf
    sys.path.insert(0, ref_path)
    from utils.convolutions import ContinuousConv          # noqa: E402  (the reference's layer)
    from utils.tools.losses import get_window_func, grid_pos  # noqa: E402
    # ---- ASCC layer (models/sym_net.py:42-53 builds it with these arguments)
    n, cin, radius = 1500, 8, 0.3
    pos = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    feat = np.maximum(rng.normal(size=(n, cin)), 0).astype(np.float32)
    conv = ContinuousConv(filters=3, kernel_size=[6, 6, 6], activation=None, use_bias=False, align_corners=True,
                          coordinate_mapping="ball_to_cube_volume_preserving", interpolation="linear", normalize=False,
                          window_function=get_window_func("peak"), radius_search_ignore_query_points=True, symmetric=True,
                          sym_axis=1, kernel_initializer=tf.keras.initializers.RandomUniform(-1.0, 1.0, seed=7))
    y = conv(feat, pos, pos, 2 * radius, None)
    out["ascc_pos"], out["ascc_feat"], out["ascc_radius"] = pos, feat, np.float32(radius)
    out["ascc_kernel"] = conv.kernel.numpy()  # the stored half kernel [6, 3, 6, cin, 3]
    out["ascc_out"] = y.numpy()
    # ---- grid_pos
    cloud = rng.uniform(0, 2, size=(5000, 3)).astype(np.float32)
    out["gridpos_cloud"] = cloud
    for stride in (2, 4):
        for central in (False, True):
            g = grid_pos(tf.constant(cloud), tf.constant([0.025 * stride] * 3, tf.float32), centralize=central)
            out[f"gridpos_s{stride}_c{int(central)}"] = g.numpy()
    # ---- rollout: run_sample.py's loop (:142-185) without inflow, 10 steps, Liquid3d checkpoint, canyon scene
    import models
    from o3d.utils import Config
    from datasets.dataset_reader_physics import read_data  # noqa: E402
    cfg = Config.load_from_file(os.path.join(ref_path, "configs", "Liquid3d.yml"))
    model = getattr(models, cfg.model.name)(**cfg.model)
    data = read_data(os.path.join(ref_path, "datasets", "canyon_data", "canyon.msgpack.zst"))[0]
    pos0 = np.asarray(data["pos"], np.float32)
    box = np.asarray(data["box"], np.float32)
    keep = np.all((box >= pos0.min(0) - 1.0) & (box <= pos0.max(0) + 1.0), axis=1)  # the crop of tests/golden/canyon_crop
    inputs = [pos0, np.asarray(data["vel"], np.float32), None, None, box[keep], np.asarray(data["box_normals"], np.float32)[keep]]
    model(inputs, training=False)  # builds the variables
    tf.train.Checkpoint(model=model).restore(os.path.join(ref_path, "checkpoints", "Liquid3d", "ckpt")).expect_partial()
    frames, times = [pos0], []
    for _ in range(10):
        t0 = time.time()
        pos, vel = model(inputs, training=False)
        times.append(time.time() - t0)
        inputs = [pos.numpy(), vel.numpy()] + inputs[2:]
        frames.append(inputs[0])
    out["rollout_pos"] = np.stack(frames)
    out["rollout_vel_last"] = inputs[1]
    out["rollout_box"], out["rollout_box_normals"] = inputs[4], inputs[5]
    out["rollout_vel0"] = np.asarray(data["vel"], np.float32)
    out["rollout_seconds_per_step"] = np.asarray(times, np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=None, help="path to a tum-pbs/DMCF checkout: also capture ASCC / grid_pos / a rollout")
    ap.add_argument("--manifest", action="store_true", help="print what a capture holds (JSON) and exit; needs numpy only")
    ap.add_argument("--inputs-only", default=None, metavar="NPZ", help="write the disputed inputs (no outputs) and exit; numpy only")
    args = ap.parse_args()
    if args.manifest:
        import json
        print(json.dumps(manifest(), indent=1))
        return
    if args.inputs_only:
        np.savez_compressed(args.inputs_only, **{f"edge_{case}_{k}": np.asarray(v) for case, c in edge_inputs().items() for k, v in c.items()})
        return
    import open3d.ml.tf as ml3d
    import tensorflow as tf

    rng = np.random.default_rng(0)
    out = {}
    cases = [("3d", 3, (4, 4, 4), 0.3, 800, 500), ("2d", 2, (1, 8, 8), 0.12, 900, 600), ("1d", 1, (1, 8, 1), 0.2, 300, 300)]
    for name, dim, ks, radius, n, m in cases:
        pts = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
        qs = rng.uniform(-1, 1, size=(m, 3)).astype(np.float32)
        if dim <= 2:
            pts[:, 2] = 0
            qs[:, 2] = 0
        if dim == 1:
            pts[:, 0] = 0
            qs[:, 0] = 0
        for ignore in (False, True):
            q = pts[:m] if ignore else qs
            frs = ml3d.layers.FixedRadiusSearch(metric="L2", ignore_query_point=ignore, return_distances=True)
            res = frs(pts, q, radius)
            tag = f"{name}_ign{int(ignore)}"
            out[f"frs_{tag}_points"], out[f"frs_{tag}_queries"], out[f"frs_{tag}_radius"] = pts, q, np.float32(radius)
            out[f"frs_{tag}_index"] = res.neighbors_index.numpy()
            out[f"frs_{tag}_row_splits"] = res.neighbors_row_splits.numpy()
            out[f"frs_{tag}_distance"] = res.neighbors_distance.numpy()
        cin, cout = 6, 5
        feat = rng.normal(size=(n, cin)).astype(np.float32)
        filt = rng.uniform(-1, 1, size=(*ks, cin, cout)).astype(np.float32)
        frs = ml3d.layers.FixedRadiusSearch(metric="L2", ignore_query_point=False, return_distances=True)
        res = frs(pts, qs, radius)
        imp = tf.clip_by_value((1 - res.neighbors_distance / radius ** 2) ** 3, 0, 1)
        for mapping in ("ball_to_cube_volume_preserving", "ball_to_cube_radial", "identity"):
            y = ml3d.ops.continuous_conv(filters=filt, out_positions=qs, extents=tf.constant([[2 * radius]], tf.float32),
                                         offset=tf.zeros((3,)), inp_positions=pts, inp_features=feat,
                                         inp_importance=tf.ones((0,), tf.float32), neighbors_index=res.neighbors_index,
                                         neighbors_row_splits=res.neighbors_row_splits, neighbors_importance=imp,
                                         align_corners=True, coordinate_mapping=mapping, interpolation="linear",
                                         normalize=False)
            out[f"cconv_{name}_{mapping}"] = y.numpy()
        out[f"cconv_{name}_feat"], out[f"cconv_{name}_filt"] = feat, filt
        out[f"cconv_{name}_index"] = res.neighbors_index.numpy()
        out[f"cconv_{name}_row_splits"] = res.neighbors_row_splits.numpy()
        out[f"cconv_{name}_importance"] = imp.numpy()
    v = rng.normal(size=100).astype(np.float32)
    rs = np.array([0, 10, 10, 55, 100], dtype=np.int64)
    out["rss_values"], out["rss_row_splits"] = v, rs
    out["rss_out"] = ml3d.ops.reduce_subarrays_sum(v, rs).numpy()
    capture_edges(out)
    if args.reference:
        capture_reference(os.path.abspath(args.reference), out, np.random.default_rng(1))
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "open3d_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, len(out), "arrays")


if __name__ == "__main__":
    main()
