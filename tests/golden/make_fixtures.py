"""Generates the committed fixtures under tests/golden/ from data files the reference holds.
Run in the build container (needs /root/reference); the GPU box only sees the committed outputs.

  liquid3d_weights.npz   the 18 CConv kernels/biases + 7 Dense layers of checkpoints/Liquid3d
                         (tensor-bundle decoded by dmcf_amd/utils/tf_checkpoint.py; keys = reference
                         checkpoint variable paths)
  ckpt_shapes.json       variable path -> shape for all three shipped checkpoints (WaterRamps / WBC-SPH
                         have an index but no data blob, .MISSING_LARGE_BLOBS)
  canyon_crop.msgpack.zst  the first 3 of the 13 frames of datasets/canyon_data/canyon.msgpack.zst (the scene of
                         run_sample.py) with the static boundary cropped to the fluid's bounding box +- 1.0
                         (10,006 of 185,447 boundary particles), re-encoded in the same file format by
                         dmcf_amd.datasets.write_scene
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dmcf_amd.utils import tf_checkpoint as tc  # noqa: E402

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    w = tc.load_checkpoint(os.path.join(REF, "checkpoints/Liquid3d/ckpt"))
    np.savez_compressed(os.path.join(OUT, "liquid3d_weights.npz"), **{k: v for k, v in w.items() if k.startswith("model/")})
    shapes = {}
    for name in ("Liquid3d", "WaterRamps", "WBC-SPH"):
        idx = tc.read_index(os.path.join(REF, f"checkpoints/{name}/ckpt.index"))
        shapes[name] = {k[:-len("/.ATTRIBUTES/VARIABLE_VALUE")]: list(e["shape"]) for k, e in sorted(idx.items())
                        if k.startswith("model/") and k.endswith("/.ATTRIBUTES/VARIABLE_VALUE")
                        and ".OPTIMIZER_SLOT" not in k}
    json.dump(shapes, open(os.path.join(OUT, "ckpt_shapes.json"), "w"), indent=0, sort_keys=True)
    print({k: len(v) for k, v in shapes.items()})
    from dmcf_amd.datasets import read_scene, write_scene  # noqa: E402
    frames = read_scene(os.path.join(REF, "datasets/canyon_data/canyon.msgpack.zst"))
    p0, box = frames[0]["pos"], frames[0]["box"]
    keep = np.all((box >= p0.min(0) - 1.0) & (box <= p0.max(0) + 1.0), axis=1)
    crop = [dict(f) for f in frames[:3]]
    crop[0]["box"] = np.ascontiguousarray(box[keep])
    crop[0]["box_normals"] = np.ascontiguousarray(frames[0]["box_normals"][keep])
    write_scene(os.path.join(OUT, "canyon_crop.msgpack.zst"), crop)
    print("canyon crop:", int(keep.sum()), "boundary particles,", len(crop), "frames")
