"""One-off diagnostic: ONE step of the bench's full-size scene (100^3 fluid + shell) on the HIP path against the CPU oracle
(about a minute and tens of GB of host memory).  Usage: python tools/check_full_size.py [side] [steps_before]
``steps_before``: HIP-path steps to advance the scene before the compared step (the compared step starts from the HIP path's
own state)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from dmcf_amd import models
from dmcf_amd.pipelines import Simulator
from dmcf_amd.utils import tf_checkpoint as tc
from oracle.model_ref import ModelRef
from tools import configs, scenes

side = int(sys.argv[1]) if len(sys.argv) > 1 else 100
before = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
cfg = configs.LIQUID3D
w = dict(np.load(os.path.join(ROOT, "tests", "golden", "liquid3d_weights.npz")))
model = getattr(models, cfg["name"])(**cfg); tc.load_into_model(model, w, device=dev)
sim = Simulator(model, device="cuda:0", reserve_gib="auto")
state = scenes.model_inputs(scenes.box_scene(side), device=dev)
for _ in range(before):
    state = sim.step([state])[0]
inp = [None if x is None else x.cpu().numpy() for x in state]
out = sim.step([state])[0]
t0 = time.time()
ref = ModelRef(cfg, w)
pos_ref, vel_ref = ref.step(inp)
print(f"oracle step: {time.time() - t0:.1f} s, {ref.pairs} pairs", flush=True)
pos = out[0].cpu().numpy()
err = np.abs(pos - pos_ref).max() / np.abs(pos_ref).max()
corr, cref = model.pos_correction.cpu().numpy(), ref.pos_correction
cerr = np.abs(corr - cref).max() / np.abs(cref).max()
print(f"side {side}, after {before} steps: pos rel err {err:.2e}, correction rel err {cerr:.2e} (f32 oracle), max |corr| {np.abs(cref).max():.3e}, "
      f"max speed {np.linalg.norm(vel_ref, axis=1).max():.2f}")
