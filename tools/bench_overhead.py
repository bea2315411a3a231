"""Per-output overhead of the CConv kernels: the L14 layer shape (32 -> 32, 1.12M outputs) with EMPTY neighbour lists
(no splat work at all: what remains is tile handling + contraction + epilogue)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmcf_amd import ops
from tools import scenes
from tools.microbench import timed  # noqa

dev = torch.device("cuda:0")
sc = scenes.box_scene(100)
s0 = torch.from_numpy(np.concatenate([sc["pos"], sc["box"]])).to(dev)
n = s0.shape[0]
g = torch.Generator(device=dev).manual_seed(0)
for cin, cout in ((32, 32), (16, 32), (24, 16), (8, 8), (4, 8)):
    feat = torch.rand(n, cin, device=dev, generator=g)
    W = torch.rand(4, 4, 4, cin, cout, device=dev, generator=g) - 0.5
    rs = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    idx = torch.zeros(1, dtype=torch.int32, device=dev)
    d = torch.zeros(1, dtype=torch.float32, device=dev)
    for k in ("", "cls", "z3", "pair"):  # "": the default dispatch of a short-row layer
        os.environ.pop("DMCF_CCONV_KERNEL", None)
        if k:
            os.environ["DMCF_CCONV_KERNEL"] = k
        f = lambda: ops.cconv_forward(W, s0, 0.2, s0, feat, idx, rs, window="poly6", row_length_hint=1)
        try:
            f()
        except Exception as e:
            print(f"{cin}->{cout} {k}: {type(e).__name__}", flush=True)
            continue
        print(f"{cin}->{cout} {k or 'default'}: {timed(f):.2f} ms with empty neighbour lists", flush=True)
