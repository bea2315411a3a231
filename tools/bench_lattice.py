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

def box(c):
    lo = c.amin(dim=0)
    return lo.tolist(), (c.amax(dim=0) - lo + 1).tolist()

def lin_of(c, lo, dim):
    d = (c - torch.tensor(lo, device=dev, dtype=torch.int32)).long()
    return (d[:, 2] * dim[1] + d[:, 1]) * dim[0] + d[:, 0]

for name, inp, out, istep, ostep, R, cin, cout in [("s1->s1  8->16", s1, s1, 1, 1, 0.2, 8, 16), ("s1->s2  8->8 ", s1, s2, 1, 2, 0.4, 8, 8),
                                                   ("s2->s2  4->8 ", s2, s2, 2, 2, 0.4, 4, 8)]:
    feat = torch.rand(inp.shape[0], cin, device=dev, generator=g)
    W = torch.rand(4, 4, 4, cin, cout, device=dev, generator=g) - 0.5
    ic, oc = cells(inp, istep), cells(out, ostep)
    ilo, idim = box(ic)
    olo, odim = box(oc)
    voxel = [h * istep] * 3
    step = ostep // istep
    ilo, idim = ops.lattice_volume_box(olo, odim, step, ops.lattice_reach(voxel, R, dev), ilo, idim)  # padded
    il, ol = lin_of(ic, ilo, idim), lin_of(oc, olo, odim)
    def prep():
        vol = feat.new_zeros((idim[2] * idim[1] * idim[0], cin))
        vol[il] = feat
        tab = torch.full((odim[2] * odim[1] * odim[0],), -1, dtype=torch.int32, device=dev)
        tab[ol] = torch.arange(oc.shape[0], dtype=torch.int32, device=dev)
        return vol.view(idim[2], idim[1], idim[0], cin), tab.view(odim[2], odim[1], odim[0])
    t_tab = timed(prep)
    vol, tab = prep()
    voxel = [h * istep] * 3
    f_lat = lambda: ops.lattice_conv(W, vol, ilo, tab, olo, out.shape[0], voxel, 2 * R, inp_step=ostep // istep, window="poly6")
    y = f_lat()
    t_lat = timed(f_lat)
    t_search = timed(lambda: ops.fixed_radius_search(inp, out, R, return_distances=True))
    nns = ops.fixed_radius_search(inp, out, R, return_distances=True)
    f_nl = lambda: ops.cconv_forward(W, out, 2 * R, inp, feat, nns.neighbors_index, nns.neighbors_row_splits,
                                     neighbors_value=nns.neighbors_distance, window="poly6")
    z = f_nl()
    t_nl = timed(f_nl)
    err = float((y - z).abs().max() / z.abs().max())
    print(f"{name}: pairs {nns.neighbors_index.shape[0]/1e6:6.1f}M  lattice {t_lat:6.2f} ms (+ volume / table {t_tab:5.2f})   neighbour list: "
          f"conv {t_nl:6.2f} ms + search {t_search:5.2f} ms   max diff {err:.1e}", flush=True)
