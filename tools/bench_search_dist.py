"""Padded (single-pass) search with and without the distance output on the bench scene's lists."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmcf_amd import ops
from dmcf_amd.utils.tools.losses import grid_pos
from tools import scenes
from tools.bench_search import timed, s0, s1, s2

for name, pts, qs, R in [("s0->s0 R0.1", s0, s0, 0.1), ("s0->s1 R0.2", s0, s1, 0.2), ("s1->s0 R0.2", s1, s0, 0.2),
                         ("s0->s2 R0.4", s0, s2, 0.4), ("s2->s0 R0.4", s2, s0, 0.4)]:
    table = ops.build_spatial_hash_table(pts, R, n_queries=qs.shape[0])
    r = ops.fixed_radius_search(pts, qs, R, hash_table=table)
    total = r.neighbors_index.shape[0]
    for rd in (True, False):
        t = timed(lambda: ops.fixed_radius_search(pts, qs, R, hash_table=table, capacity_hint=total, return_distances=rd))
        print(f"{name}: {total/1e6:7.1f}M pairs  return_distances={rd}: {t:6.2f} ms", flush=True)
