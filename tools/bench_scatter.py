"""Splat S against the gather kernels on the bench's own geometry: the particles of the 100^3 box (+ shell) onto the coarse
grid_pos lattices, 24 -> 4 / 16 -> 8 (s2, voxel 0.1, radius 0.4) and 24 -> 8 (s1, voxel 0.05, radius 0.2).
    python tools/bench_scatter.py [side]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmcf_amd import ops  # noqa: E402
from tools import scenes  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return min(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    dev = torch.device("cuda:0")
    sc = scenes.box_scene(side)
    P = torch.from_numpy(np.concatenate([sc["pos"], sc["box"]])).to(dev)
    g = torch.Generator().manual_seed(0)
    for name, voxel, radius, layers, blocks in (("s0->s2", 0.1, 0.4, ((24, 4), (16, 8)), (2, 3, 4)), ("s0->s1", 0.05, 0.2, ((24, 8),), (2,))):
        Q = ops.grid_pos(P, torch.tensor([voxel] * 3), centralize=True)
        fwd = ops.fixed_radius_search(P, Q, radius, return_distances=False, row_stride=int(2700 if radius > 0.3 else 420))
        t = ops.fixed_radius_search(Q, P, radius, return_distances=False, row_stride=420)
        pairs = int(fwd.row_count.sum())
        print(f"{name}: {P.shape[0]} inputs, {Q.shape[0]} outputs, {pairs:.4g} pairs (transposed {int(t.row_count.sum()):.4g}), "
              f"longest rows {int(fwd.max_count)} / {int(t.max_count)}")
        for cin, cout in layers:
            F = torch.relu(torch.randn(P.shape[0], cin, generator=g)).to(dev)
            W = (torch.rand(4, 4, 4, cin, cout, generator=g) - 0.5).to(dev)
            kname = ops.cconv_forward(W, Q, 2 * radius, P, F, fwd.raw()[0], fwd.raw()[1], window="poly6", neighbors_row_count=fwd.row_count,
                                      row_length_hint=2, name_only=True)
            yg = ops.cconv_forward(W, Q, 2 * radius, P, F, fwd.raw()[0], fwd.raw()[1], window="poly6", neighbors_row_count=fwd.row_count, row_length_hint=2)
            ms_g = timed(lambda: ops.cconv_forward(W, Q, 2 * radius, P, F, fwd.raw()[0], fwd.raw()[1], window="poly6",
                                                   neighbors_row_count=fwd.row_count, row_length_hint=2))
            print(f"  {cin:2d} -> {cout}: gather {kname}: {ms_g:.3f} ms")
            for m in blocks:
                try:
                    ms_plan = timed(lambda: ops.scatter_plan(P, Q, voxel, radius, block_cells=m))
                    plan = ops.scatter_plan(P, Q, voxel, radius, block_cells=m)
                    flag = torch.zeros(1, dtype=torch.int32, device=dev)
                    ys = ops.cconv_scatter_forward(W, Q, 2 * radius, P, F, t.raw()[0], t.raw()[1], t.row_count, plan, window="poly6", error_flag=flag)
                    ms_s = timed(lambda: ops.cconv_scatter_forward(W, Q, 2 * radius, P, F, t.raw()[0], t.raw()[1], t.row_count, plan, window="poly6"))
                    err = float((ys - yg).abs().max() / yg.abs().max())
                    hdr = plan.buf[:256].view(torch.int32).cpu().numpy()
                    print(f"           scatter m={m}: {ms_s:.3f} ms (+ plan {ms_plan:.3f} once per point-set pair); {hdr[11]} blocks of a {hdr[7]}x{hdr[8]}x{hdr[9]} region, {hdr[12]} overflow rows; "
                          f"max |S - gather| / max |gather| = {err:.2e}; flag {int(flag)}")
                except Exception as e:
                    print(f"           scatter m={m}: {type(e).__name__}: {e}")


if __name__ == "__main__":
    main()
