"""Debug driver for cconv_ws.hip: one small forced launch against the cls kernel."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmcf_amd import ops

def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    n, m = int(os.environ.get("N", 3000)), int(os.environ.get("M", 1000))
    cin, cout = int(os.environ.get("CIN", 32)), int(os.environ.get("COUT", 32))
    R = float(os.environ.get("R", 0.12))
    inp = torch.rand(n, 3, generator=g).to(dev)
    out = torch.rand(m, 3, generator=g).to(dev)
    feat = torch.randn(n, cin, generator=g).to(dev)
    W = (torch.rand(4, 4, 4, cin, cout, generator=g) - 0.5).to(dev)
    nns = ops.fixed_radius_search(inp, out, R, return_distances=False)
    print("pairs", nns.neighbors_index.shape[0], "longest", int(torch.diff(nns.neighbors_row_splits).max()), flush=True)
    res = {}
    for k in ("cls", "ws"):
        os.environ["DMCF_CCONV_KERNEL"] = k
        y = ops.cconv_forward(W, out, 2 * R, inp, feat, nns.neighbors_index, nns.neighbors_row_splits, window="poly6")
        torch.cuda.synchronize()
        res[k] = y
        print(k, "ok", float(y.abs().max()), flush=True)
    d = (res["ws"] - res["cls"]).abs().max().item() / res["cls"].abs().max().item()
    print("max rel diff ws vs cls", d)
    bad = ((res["ws"] - res["cls"]).abs() > 1e-4 * res["cls"].abs().max()).any(1).nonzero().flatten()
    print("bad rows", bad.numel(), bad[:40].tolist())

if __name__ == "__main__":
    main()
