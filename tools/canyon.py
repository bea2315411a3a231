"""One model step against physical ground truth: the frames of the reference's canyon scene.

tests/golden/canyon_frames.npz (made by tests/golden/make_canyon_frames.py) holds frames 7 ... 12 of
datasets/canyon_data/canyon.msgpack.zst -- the scene run_sample.py:160-179 steps with the shipped Liquid3d
checkpoint, i.e. data of the kind those weights were trained to continue.  One step from frame t must land closer
to frame t + 1 than the bare integration (models/pbf_model.py:234-250) does, and the network's correction must
point along the correction the data asks for.  Used by tests/test_canyon_frames.py (oracle on CPU, HIP path on the
GPU) and tools/reading_sweep.py (the same numbers under every disputed reading of the absent Open3D library).
"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "canyon_frames.npz")
WEIGHTS = os.path.join(ROOT, "tests", "golden", "liquid3d_weights.npz")

# bars of VERDICT r05 item 1 (measured on the oracle: ratio 0.64 - 0.75, cosine 0.79 - 0.81 for t = 8 ... 11)
RATIO_BAR = 0.8
COSINE_BAR = 0.7
FRAMES = (8, 9, 10, 11)


def load():
    d = dict(np.load(FIXTURE))
    d["first"] = int(d["frame_id"][0])
    return d


def inputs(fx, t):
    """Model inputs [pos, vel, acc, feats, box, box_normals] of frame t (numpy float32)."""
    i = t - fx["first"]
    return [fx["pos"][i], fx["vel"][i], None, None, fx["box"], fx["box_normals"]]


def integrate(pos, vel, timestep=0.02, grav=-9.81):
    """models/pbf_model.py:234-240 in float32."""
    f32 = np.float32
    vel2 = (vel + f32(timestep) * np.array([0, grav, 0], dtype=f32)).astype(f32)
    return (pos + f32(timestep) * vel2).astype(f32)


def score(fx, t, pos_pred):
    """(ratio, cosine): mean |pred - frame t+1| over mean |integration - frame t+1|, and the cosine between the
    network's correction (pred - integration) and the needed one (frame t+1 - integration), all particles stacked."""
    i = t - fx["first"]
    target = fx["pos"][i + 1].astype(np.float64)
    pint = integrate(fx["pos"][i], fx["vel"][i]).astype(np.float64)
    pred = np.asarray(pos_pred, dtype=np.float64)
    e_net = np.linalg.norm(pred - target, axis=1).mean()
    e_int = np.linalg.norm(pint - target, axis=1).mean()
    c, n = (pred - pint).ravel(), (target - pint).ravel()
    cos = float(c @ n / max(np.linalg.norm(c) * np.linalg.norm(n), 1e-300))
    return float(e_net / e_int), cos


ROLLOUT_FROM, ROLLOUT_STEPS, ROLLOUT_BAR = 8, 4, 0.4


def rollout_score(fx, step, t0=ROLLOUT_FROM, k=ROLLOUT_STEPS):
    """A free-running rollout of k steps from frame t0 against frame t0 + k: mean |rollout - frame| over mean |k bare integration
    steps - frame|.  ``step(pos, vel) -> (pos', vel')`` on numpy float32.  Sharper than one step (errors of a wrong reading compound,
    the network's help accumulates): the oracle reaches 0.30 -- the network removes 70 % of what integration alone misses -- and e.g.
    ``out_scale`` x 0.5, which one step cannot tell from the shipped value, reads 0.47."""
    i = t0 - fx["first"]
    pos, vel = fx["pos"][i], fx["vel"][i]
    pi, vi = pos.copy(), vel.copy()
    g = np.array([0, -9.81, 0], dtype=np.float32)
    for _ in range(k):
        pos, vel = step(pos, vel)
        vi = (vi + np.float32(0.02) * g).astype(np.float32)
        pi = (pi + np.float32(0.02) * vi).astype(np.float32)
    target = fx["pos"][i + k].astype(np.float64)
    e = np.linalg.norm(np.asarray(pos, dtype=np.float64) - target, axis=1).mean()
    e_int = np.linalg.norm(pi.astype(np.float64) - target, axis=1).mean()
    return float(e / e_int)
