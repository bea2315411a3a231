#!/usr/bin/env python
"""Capture TRUE golden vectors from the reference's own operator library.

Run this ONCE on any machine with ``tensorflow==2.5`` and ``open3d==0.15.2`` (the reference's
requirements.txt); it does not need the reference repository.  It writes ``tests/golden/open3d_golden.npz``
with inputs and outputs of the three Open3D operators on the DMCF hot path for the flag sets DMCF uses.
Commit the file: ``tests/test_oracle.py::test_against_open3d_golden`` then pins the oracle (and through it
the HIP path) to the real library and the "parity unpinned" caveat in DESIGN.md can be dropped.

Nothing in this script can run in the build container or on the GPU box (neither package exists for
ROCm 7 / Python 3.10); it is provided so the gap can be closed off-box.
"""
import os

import numpy as np


def main():
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
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "open3d_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, len(out), "arrays")


if __name__ == "__main__":
    main()
