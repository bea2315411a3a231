"""Time the exact (host round trip) and the estimated-size search on the neighbour lists of the 1M-particle bench scene.
A single-pass variant (count + decoupled look-back scan + write in one kernel, hits parked in LDS) was measured with
this script and dropped: 4.0 ms against 3.8 ms for the 307M-pair lists, 2.4 against 1.2 ms for the 33M-pair one."""
import os, sys
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

dev = torch.device("cuda:0")
sc = scenes.box_scene(int(os.environ.get("SIDE", "100")))
s0 = torch.from_numpy(np.concatenate([sc["pos"], sc["box"]])).to(dev)
s1 = grid_pos(s0, np.float32([0.05] * 3), centralize=True)
s2 = grid_pos(s0, np.float32([0.1] * 3), centralize=True)
for name, pts, qs, R in [("s0->s0 R0.1", s0, s0, 0.1), ("s0->s1 R0.2", s0, s1, 0.2), ("s1->s0 R0.2", s1, s0, 0.2),
                         ("s0->s2 R0.4", s0, s2, 0.4), ("s2->s0 R0.4", s2, s0, 0.4), ("s2->s2 R0.4", s2, s2, 0.4)]:
    table = ops.build_spatial_hash_table(pts, R, n_queries=qs.shape[0])
    r = ops.fixed_radius_search(pts, qs, R, hash_table=table)
    total = r.neighbors_index.shape[0]
    t2 = timed(lambda: ops.fixed_radius_search(pts, qs, R, hash_table=table))
    t1 = timed(lambda: ops.fixed_radius_search(pts, qs, R, hash_table=table, capacity_hint=total))
    # the rollout's form: ONE pass into padded rows (stride from the longest row, as the per-step cache would choose it), index only
    from dmcf_amd.utils.convolutions import row_stride
    stride = row_stride(int(torch.diff(r.neighbors_row_splits).max()))
    pad = lambda: ops.fixed_radius_search(pts, qs, R, hash_table=table, row_stride=stride, return_distances=False)
    pad()
    t0 = timed(pad)
    print(f"{name}: {total/1e6:7.1f}M pairs  exact {t2:6.2f} ms (incl. host round trip)  estimated {t1:6.2f} ms  padded {t0:6.2f} ms "
          f"({qs.shape[0]} queries, stride {stride})", flush=True)
