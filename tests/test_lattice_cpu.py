"""Host logic of the lattice form of ContinuousConv (dmcf_amd/lattice.py): registry, family / ratio rules, cells, table,
volume.  The kernel itself is covered by tests/test_gpu_ops.py::test_lattice_conv_matches_neighbour_list_form."""
import numpy as np
import torch

from dmcf_amd import lattice


def _points(cells, voxel, center):
    return (torch.tensor(cells, dtype=torch.float32) * torch.tensor(voxel, dtype=torch.float32)
            + torch.tensor(center, dtype=torch.float32)).contiguous()


def test_cells_table_volume_roundtrip():
    rng = np.random.default_rng(0)
    cells = np.unique(rng.integers(-4, 7, size=(200, 3)), axis=0).astype(np.int32)
    center, voxel = [0.37, -1.2, 2.5], [0.05, 0.05, 0.05]
    pos = _points(cells, voxel, center)
    lo = cells.min(axis=0) - 1
    info = lattice.LatticeInfo(pos, torch.tensor(center), voxel, "f", lo, cells.max(axis=0) - lo + 2)
    assert np.array_equal(info.cells().numpy(), cells)
    t = info.table()
    assert t.shape == (info.dims[2], info.dims[1], info.dims[0]) and int((t >= 0).sum()) == cells.shape[0]
    d = cells - lo
    assert np.array_equal(t[d[:, 2], d[:, 1], d[:, 0]].numpy(), np.arange(cells.shape[0]))
    feats = torch.arange(cells.shape[0] * 3, dtype=torch.float32).reshape(-1, 3)
    v = info.volume(feats)
    assert torch.equal(v[d[:, 2], d[:, 1], d[:, 0]], feats) and float(v.abs().sum()) == float(feats.abs().sum())


def test_pair_rules(monkeypatch):
    lattice.clear()
    center = torch.tensor([0.0, 0.0, 0.0])
    g = np.stack(np.meshgrid(np.arange(4), np.arange(4), np.arange(4), indexing="ij"), -1).reshape(-1, 3).astype(np.int32)
    fine = _points(g, [0.05] * 3, [0, 0, 0])
    coarse = _points(g[:20], [0.1] * 3, [0, 0, 0])
    other = _points(g[:30], [0.05] * 3, [0, 0, 0])
    third = _points(g[:25], [0.2] * 3, [0, 0, 0])
    odd = _points(g[:26], [0.15] * 3, [0, 0, 0])
    lattice.register(fine, center, [0.05] * 3, "A", [0, 0, 0], [4, 4, 4])
    lattice.register(coarse, center, [0.1] * 3, "A", [0, 0, 0], [4, 4, 4])
    lattice.register(other, center, [0.05] * 3, "B", [0, 0, 0], [4, 4, 4])
    lattice.register(third, center, [0.2] * 3, "A", [0, 0, 0], [4, 4, 4])
    lattice.register(odd, center, [0.15] * 3, "A", [0, 0, 0], [4, 4, 4])
    assert lattice.pair(fine, fine).ratio == 1
    assert lattice.pair(fine, coarse).ratio == 2       # outputs on the coarser lattice
    assert lattice.pair(coarse, fine).ratio == 0.5     # outputs on the finer lattice: eight parity launches
    assert lattice.pair(fine, third).ratio == 4
    assert lattice.pair(third, fine) is None           # 1/4: not supported
    assert lattice.pair(fine, odd) is None             # fl(0.15) is not 3 * fl(0.05): the lattices do not coincide exactly
    assert lattice.pair(coarse, odd) is None           # 1.5
    assert lattice.pair(fine, other) is None           # different families never share a centre
    assert lattice.pair(fine, fine.clone()) is None    # not a registered tensor
    fine.add_(0.0)                                     # modified in place after registration
    assert lattice.pair(fine, coarse) is None
    monkeypatch.setenv("DMCF_LATTICE_CONV", "0")
    assert lattice.pair(coarse, coarse) is None
    monkeypatch.delenv("DMCF_LATTICE_CONV")
    assert lattice.pair(coarse, coarse).ratio == 1
    lattice.clear()
    assert lattice.pair(coarse, coarse) is None
    # collapsed axes (2-D scenes) are not registered
    lattice.register(coarse, center, [0.1, 0.1, 0.0], "A", [0, 0, 0], [4, 4, 1])
    assert lattice.lookup(coarse) is None


def test_core_box_follows_the_dense_slabs():
    """lattice._pick_core: per axis the run of slabs around the fullest one that hold at least CORE_SLAB_FRACTION of its points;
    stray cells far out do not widen it -- not even a well filled slab beyond a gap --, an empty histogram keeps the whole box."""
    from dmcf_amd import lattice
    hx = [0, 1, 0, 0, 90, 100, 95, 80, 3, 0, 0, 1]
    hy = [2, 50, 60, 2, 55, 1, 0, 0, 0, 40]
    hz = [0, 0, 0]
    lo, hi = lattice._pick_core([-4, 10, 0], [12, 10, 3], [hx, hy, hz])
    assert lo == [0, 11, 0] and hi == [4, 15, 3]
