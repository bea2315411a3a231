"""Debug helper: cconv_cls.hip against cconv_blk.hip row by row."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmcf_amd import ops

def run(kernel, *a, **k):
    os.environ["DMCF_CCONV_KERNEL"] = kernel
    return ops.cconv_forward(*a, **k)

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for n, m, cin, cout, radius in [(50, 8, 16, 16, 0.9), (400, 40, 16, 16, 0.5), (3000, 64, 16, 16, 0.4), (3000, 64, 8, 16, 0.4), (3000, 64, 32, 16, 0.4)]:
    inp = torch.tensor(rng.uniform(0, 1, size=(n, 3)).astype(np.float32), device=dev)
    out = torch.tensor(rng.uniform(0, 1, size=(m, 3)).astype(np.float32), device=dev)
    feat = torch.tensor(rng.normal(size=(n, cin)).astype(np.float32), device=dev)
    W = torch.tensor(rng.uniform(-1, 1, size=(4, 4, 4, cin, cout)).astype(np.float32), device=dev)
    nns = ops.fixed_radius_search(inp, out, radius, return_distances=True)
    args = (W, out, 2 * radius, inp, feat, nns.neighbors_index, nns.neighbors_row_splits)
    kw = dict(neighbors_value=nns.neighbors_distance, window="poly6")
    a = run("blk", *args, **kw)
    b = run("cls", *args, **kw)
    cnt = torch.diff(nns.neighbors_row_splits).cpu().numpy()
    err = ((a - b).abs().amax(dim=1) / a.abs().amax()).cpu().numpy()
    print(f"n={n} m={m} cin={cin}: max err {err.max():.2e}")
    for r in range(min(m, 24)):
        print(f"   row {r:3d} count {cnt[r]:4d} err {err[r]:.2e}")
