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

Nothing in this script can run in the build container or on the GPU box (neither package exists for
ROCm 7 / Python 3.10); it is provided so the gap can be closed off-box.
"""
import argparse
import os
import sys
import time

import numpy as np


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
    args = ap.parse_args()
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
    if args.reference:
        capture_reference(os.path.abspath(args.reference), out, np.random.default_rng(1))
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "open3d_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, len(out), "arrays")


if __name__ == "__main__":
    main()
