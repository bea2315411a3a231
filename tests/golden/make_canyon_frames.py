"""Generates tests/golden/canyon_frames.npz from the reference's own scene file (data only, no reference source).

/root/reference/datasets/canyon_data/canyon.msgpack.zst holds 13 consecutive frames (dt = 0.02) of the SPH simulation
the shipped Liquid3d checkpoint was trained to continue (run_sample.py:160-179 steps exactly this scene with exactly
these weights).  That makes the pair (frame t, frame t + 1) physical ground truth for ONE model step: a restatement of
the operators with a wrong filter layout, interpolation or window cannot land closer to frame t + 1 than the bare
integration does (tests/test_canyon_frames.py, tools/reading_sweep.py).

  pos, vel   [6, 1280, 3] float32   frames 7 ... 12 (the column reaches the canyon floor in frame 7)
  frame_id   [6] int64
  box, box_normals  [M, 3] float32  the static boundary within the fluid's bounding box over those frames +- 1.0
                                    (a superset of the model's own crop, +- 0.8 = the largest filter extent,
                                    models/pbf_model.py:330-336, so a step sees what it would see in the full scene)

Run in the build container: python tests/golden/make_canyon_frames.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dmcf_amd.datasets import read_scene  # noqa: E402

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
FIRST, LAST = 7, 12

if __name__ == "__main__":
    frames = read_scene(os.path.join(REF, "datasets/canyon_data/canyon.msgpack.zst"))
    sel = frames[FIRST:LAST + 1]
    pos = np.stack([f["pos"] for f in sel]).astype(np.float32)
    vel = np.stack([f["vel"] for f in sel]).astype(np.float32)
    box, nrm = frames[0]["box"], frames[0]["box_normals"]
    lo, hi = pos.reshape(-1, 3).min(0), pos.reshape(-1, 3).max(0)
    keep = np.all((box >= lo - 1.0) & (box <= hi + 1.0), axis=1)
    np.savez_compressed(os.path.join(OUT, "canyon_frames.npz"), pos=pos, vel=vel,
                        frame_id=np.array([int(f["frame_id"]) for f in sel], dtype=np.int64),
                        box=np.ascontiguousarray(box[keep], dtype=np.float32),
                        box_normals=np.ascontiguousarray(nrm[keep], dtype=np.float32))
    print("frames", FIRST, "...", LAST, pos.shape, "boundary", int(keep.sum()), "of", len(box))
