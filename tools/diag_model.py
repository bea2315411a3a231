"""Diagnostic: stage-by-stage error of the HIP model step against the oracle (f32 and f64 operators)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.model_ref import ModelRef
from dmcf_amd import models
from dmcf_amd.utils import tf_checkpoint as tc
from dmcf_amd.utils.convolutions import neighbor_cache
from tools import configs, scenes

dev = torch.device("cuda:0")
w = dict(np.load(os.path.join(ROOT, "tests/golden/liquid3d_weights.npz")))
cfg = configs.LIQUID3D
scene = scenes.box_scene(12)
model = getattr(models, cfg["name"])(**cfg)
tc.load_into_model(model, w, device=dev)
r32, r64 = ModelRef(cfg, w), ModelRef(cfg, w, f64=True)
dn = scenes.model_inputs(scene)
dt_ = scenes.model_inputs(scene, device=dev)
p32, v32 = r32.step(dn)
p64, v64 = r64.step(dn)
with neighbor_cache():
    d = model.transform(dt_)
    x = model.preprocess(d)
    out = model.run_forward(x, d)
    res = model.postprocess(out, d)
rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
o = out.cpu().numpy()
print("net_output   hip-vs-64 %.2e  ref32-vs-64 %.2e  |out|max %.3g" % (rel(o, r64.net_output), rel(r32.net_output, r64.net_output), np.abs(r64.net_output).max()))
print("dilated sizes", [p.shape[0] for p in x[0]], "ref", "n/a")
print("pos          hip-vs-64 %.2e  ref32-vs-64 %.2e" % (rel(res[0].cpu().numpy(), p64), rel(p32, p64)))
print("vel          hip-vs-64 %.2e  ref32-vs-64 %.2e" % (rel(res[1].cpu().numpy(), v64), rel(v32, v64)))
print("corr max", np.abs(r64.pos_correction).max(), "pos max", np.abs(p64).max(), "vel max", np.abs(v64).max())
# per-stage: feed the oracle's HRNet output into both ASCC implementations
