"""Time dmcf_lattice_conv_forward against the neighbour-list path (search + CConv kernel) on the lattice -> lattice layers
of the 1M-particle bench scene: s1 -> s1 (8 -> 16), s1 -> s2 (8 -> 8), s2 -> s2 (4 -> 8)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmcf_amd import ops
from dmcf_amd.utils.tools.losses import grid_pos
from tools import scenes
from tools.microbench import timed

dev = torch.device("cuda:0")
sc = scenes.box_scene(int(os.environ.get("SIDE", "100")))
s0 = torch.from_numpy(np.concatenate([sc["pos"], sc["box"]])).to(dev)
h = 0.05
s1 = grid_pos(s0, np.float32([h] * 3), centralize=True)
s2 = grid_pos(s0, np.float32([2 * h] * 3), centralize=True)
ref = s2[0]
g = torch.Generator(device=dev).manual_seed(0)

def cells(x, step):
    c = torch.round((x - ref) / np.float32(h)).to(torch.int32)
    assert torch.all(c % step == 0)
    return (c // step).contiguous()

def table_of(c):
    tmin = c.amin(dim=0)
    tdim = (c.amax(dim=0) - tmin + 1).tolist()
    t = torch.full((tdim[2], tdim[1], tdim[0]), -1, dtype=torch.int32, device=dev)
    d = (c - tmin).long()
    t[d[:, 2], d[:, 1], d[:, 0]] = torch.arange(c.shape[0], dtype=torch.int32, device=dev)
    return t, tmin.tolist()

for name, inp, out, istep, ostep, R, cin, cout in [("s1->s1  8->16", s1, s1, 1, 1, 0.2, 8, 16), ("s1->s2  8->8 ", s1, s2, 1, 2, 0.4, 8, 8),
                                                   ("s2->s2  4->8 ", s2, s2, 2, 2, 0.4, 4, 8)]:
    feat = torch.rand(inp.shape[0], cin, device=dev, generator=g)
    W = torch.rand(4, 4, 4, cin, cout, device=dev, generator=g) - 0.5
    ic, oc = cells(inp, istep), cells(out, ostep)
    t_tab = timed(lambda: table_of(ic))
    tab, tmin = table_of(ic)
    voxel = [h * istep] * 3
    f_lat = lambda: ops.lattice_conv(W, oc, ostep // istep, tab, tmin, voxel, 2 * R, feat, window="poly6")
    y = f_lat()
    t_lat = timed(f_lat)
    t_search = timed(lambda: ops.fixed_radius_search(inp, out, R, return_distances=True))
    nns = ops.fixed_radius_search(inp, out, R, return_distances=True)
    f_nl = lambda: ops.cconv_forward(W, out, 2 * R, inp, feat, nns.neighbors_index, nns.neighbors_row_splits,
                                     neighbors_value=nns.neighbors_distance, window="poly6")
    z = f_nl()
    t_nl = timed(f_nl)
    err = float((y - z).abs().max() / z.abs().max())
    print(f"{name}: pairs {nns.neighbors_index.shape[0]/1e6:6.1f}M  lattice {t_lat:6.2f} ms (+ table {t_tab:5.2f})   neighbour list: "
          f"conv {t_nl:6.2f} ms + search {t_search:5.2f} ms   max diff {err:.1e}", flush=True)
