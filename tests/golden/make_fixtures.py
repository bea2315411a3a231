"""Generates the committed fixtures under tests/golden/ from data files the reference holds.
Run in the build container (needs /root/reference); the GPU box only sees the committed outputs.

  liquid3d_weights.npz   the 18 CConv kernels/biases + 7 Dense layers of checkpoints/Liquid3d
                         (tensor-bundle decoded by dmcf_amd/utils/tf_checkpoint.py; keys = reference
                         checkpoint variable paths)
  ckpt_shapes.json       variable path -> shape for all three shipped checkpoints (WaterRamps / WBC-SPH
                         have an index but no data blob, .MISSING_LARGE_BLOBS)
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
