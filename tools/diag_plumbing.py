import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O
from dmcf_amd.utils.tools.losses import grid_pos
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for (n, k, m) in [(4096, 24, 16), (4096, 4, 8), (4096, 32, 32), (100000, 32, 32)]:
    x = rng.normal(size=(n, k)).astype(np.float32); w = rng.uniform(-0.3, 0.3, size=(k, m)).astype(np.float32); b = rng.normal(size=m).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64) + b
    y = torch.addmm(torch.from_numpy(b).to(dev), torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)).cpu().numpy()
    y2 = (torch.from_numpy(x).to(dev) @ torch.from_numpy(w).to(dev)).cpu().numpy() + b
    print("dense", (n, k, m), "addmm err %.2e  matmul err %.2e  numpy32 err %.2e" % (np.abs(y - ref).max() / np.abs(ref).max(), np.abs(y2 - ref).max() / np.abs(ref).max(), np.abs((x @ w + b) - ref).max() / np.abs(ref).max()))
print("allow_tf32", torch.backends.cuda.matmul.allow_tf32, "precision", torch.get_float32_matmul_precision())
from tools import scenes
s = scenes.box_scene(12)
pos = np.concatenate([s["pos"], s["box"]])
for stride in (2, 4):
    vs = np.float32([0.025] * 3) * np.float32(stride)
    g = grid_pos(torch.from_numpy(pos).to(dev), vs, centralize=True).cpu().numpy()
    r = O.grid_pos(pos, vs, centralize=True)
    print("grid", stride, g.shape, r.shape, "max diff", np.abs(g - r).max() if g.shape == r.shape else None)
