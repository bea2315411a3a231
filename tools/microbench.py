"""Kernel micro-benchmarks on the 1M box: L14 (32->32, same scale), L8 (16->16, s0->s1),
ASCC.  Prints ms per launch (median of 5) for the CConv kernel and both search passes."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmcf_amd import ops
from dmcf_amd.utils.tools.losses import grid_pos
from tools import scenes

def timed(fn, n=5):
    ts = []
    for _ in range(n):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))

def main():
    dev = torch.device("cuda:0")
    side = int(os.environ.get("SIDE", "100"))
    sc = scenes.box_scene(side)
    s0 = torch.from_numpy(np.concatenate([sc["pos"], sc["box"]])).to(dev)
    s1 = grid_pos(s0, np.float32([0.05] * 3), centralize=True)
    g = torch.Generator(device=dev).manual_seed(0)
    cases = [("L3  24->8  s0->s1 R0.2", s0, s1, 0.2, 24, 8, (4, 4, 4), "poly6", False),
             ("L14 32->32 s0->s0 R0.1", s0, s0, 0.1, 32, 32, (4, 4, 4), "poly6", False),
             ("L2  24->16 s0->s0 R0.1", s0, s0, 0.1, 24, 16, (4, 4, 4), "poly6", False),
             ("L5  16->32 s0->s0 R0.1", s0, s0, 0.1, 16, 32, (4, 4, 4), "poly6", False),
             ("IN   8->16 s0->s0 R0.1", s0, s0, 0.1, 8, 16, (4, 4, 4), "poly6", False),
             ("L8  16->16 s0->s1 R0.2", s0, s1, 0.2, 16, 16, (4, 4, 4), "poly6", False),
             ("L6   8->32 s1->s0 R0.2", s1, s0, 0.2, 8, 32, (4, 4, 4), "poly6", False),
             ("L7   4->32 s1->s0 R0.2", s1, s0, 0.2, 4, 32, (4, 4, 4), "poly6", False),
             ("LP  12->64 s2->s0 R0.4", grid_pos(s0, np.float32([0.1] * 3), centralize=True), s0, 0.4, 12, 64, (4, 4, 4), "poly6", False),
             ("L4  24->4  s0->s2 R0.4", s0, grid_pos(s0, np.float32([0.1] * 3), centralize=True), 0.4, 24, 4, (4, 4, 4), "poly6", False),
             ("ASCC 32->3 s0->s0 R0.1", s0, s0, 0.1, 32, 3, (6, 3, 6), "peak", True),
             # conv200_1 + conv300_1 of Liquid3d as ONE block-diagonal launch (models/hrnet.py, _paired_convs): 8 + 16 channels
             ("LQ  24->64 s1->s0 R0.2", s1, s0, 0.2, 24, 64, (4, 4, 4), "poly6", False)]
    cases.append(("S4  24->4  s0->s2 R0.4 (splat S: scatter form over the transposed list)", s0, cases[9][2], 0.4, 24, 4, (4, 4, 4), "poly6", False))
    for name, inp, out, R, cin, cout, ks, win, sym in cases:
        if os.environ.get('ONLY') and not any(name.startswith(o) for o in os.environ['ONLY'].split(',')): continue
        if name.startswith("S4"):
            t = ops.fixed_radius_search(out, inp, R, return_distances=False)
            plan = ops.scatter_plan(inp, out, 0.1, R)
            feat = torch.rand(inp.shape[0], cin, device=dev, generator=g)
            W = torch.rand(*ks, cin, cout, device=dev, generator=g) - 0.5
            f = lambda: ops.cconv_scatter_forward(W, out, 2 * R, inp, feat, t.neighbors_index, t.neighbors_row_splits, None, plan, window=win)
            f()
            ms = timed(f)
            P = t.neighbors_index.shape[0]
            by = P * (20 + 4 * cin) + out.shape[0] * (20 + 4 * cout) + 4 * 64 * cin * cout
            print(f"{name}: pairs {P/1e6:7.1f}M  {ms:7.2f} ms  {1e6*ms/P/((cin+7)//8):.4f} ns/pair/chunk  alg {by/ms/1e6:7.0f} GB/s ({100*by/ms/1e6/8000:.1f}% of 8 TB/s)", flush=True)
            continue
        # lists without distances, as the networks use them (the window is evaluated on distances re-formed from the positions: the
        # 'plain' instantiations of splats D / E); MB_DIST=1: with the distance array
        dist = os.environ.get('MB_DIST') == '1'
        nns = ops.fixed_radius_search(inp, out, R, ignore_query_point=sym, return_distances=dist)
        feat = torch.rand(inp.shape[0], cin, device=dev, generator=g)
        W = torch.rand(*ks, cin, cout, device=dev, generator=g) - 0.5
        mask = ops.block_diagonal_tile_mask([(0, 8, 0, 32), (8, 24, 32, 64)]) if name.startswith("LQ") else 0
        f = lambda: ops.cconv_forward(W, out, 2 * R, inp, feat, nns.neighbors_index, nns.neighbors_row_splits,
                                      neighbors_value=nns.neighbors_distance if dist else None, window=win, symmetric=sym, sym_axis=1,
                                      row_length_hint=2 if R > 0.1 else 1,  # (what models/hrnet.py tells the layer)
                                      filter_tile_mask=mask)
        f()
        ms = timed(f)
        P = nns.neighbors_index.shape[0]
        K = ks[0] * ks[1] * ks[2] * (2 if sym else 1)
        by = P * (20 + 4 * cin) + out.shape[0] * (20 + 4 * cout) + 4 * K * cin * cout
        print(f"{name}: pairs {P/1e6:7.1f}M  {ms:7.2f} ms  {1e6*ms/P/((cin+7)//8):.4f} ns/pair/chunk  alg {by/ms/1e6:7.0f} GB/s ({100*by/ms/1e6/8000:.1f}% of 8 TB/s)", flush=True)


if __name__ == "__main__":
    main()
