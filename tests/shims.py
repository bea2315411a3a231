"""TEST-ONLY backend: routes dmcf_amd.ops through the CPU oracle so that host logic which normally needs a
GPU (the sharded driver, the model plumbing) can be exercised on CPU with gloo.  Never imported by the
product; installing it is an explicit act of a test."""
import numpy as np
import torch

import oracle as O
from dmcf_amd import ops


def install(monkeypatch):
    def fixed_radius_search(points, queries, radius, ignore_query_point=False, return_distances=True, hash_table=None,
                            capacity_hint=None, row_stride=None, max_count=None):
        idx, rs, d = O.fixed_radius_search(points.numpy(), queries.numpy(), radius, ignore_query_point)
        return ops.NeighborSearchResult(torch.from_numpy(idx), torch.from_numpy(rs), torch.from_numpy(d), total=len(idx))

    def build_spatial_hash_table(points, radius, n_queries=None, **kw):
        return ops.SpatialHashTable(points, radius, None, 1 << 60)

    def cconv_forward(filters, out_positions, extent, inp_positions, inp_features, neighbors_index, neighbors_row_splits,
                      neighbors_value=None, window=None, window_fac=1.0, inp_importance=None, align_corners=True,
                      coordinate_mapping="ball_to_cube_volume_preserving", interpolation="linear", normalize=False,
                      symmetric=False, sym_axis=2, bias=None, out=None, accumulate=False, n_pairs_ref=None,
                      neighbors_row_count=None, filter_tile_mask=0, skip_self=False, name_only=False, row_length_hint=0,
                      packed_cache=None):
        assert neighbors_row_count is None and not skip_self
        if name_only:
            return "cconv_oracle"  # (never "cconv_direct_kernel": the oracle backend does not share lists)
        radius = np.float32(0.5) * np.float32(extent)
        imp = None
        if window == "explicit":
            imp = neighbors_value.numpy()
        elif window is not None:
            imp = O.window(window, neighbors_value.numpy() / (radius * radius), fac=window_fac)
        k = filters.numpy()
        f = inp_features.numpy()
        n_out = out_positions.shape[0]
        kw = dict(out_positions=out_positions.numpy(), extents=extent, inp_positions=inp_positions.numpy(),
                  neighbors_index=neighbors_index.numpy(), neighbors_row_splits=neighbors_row_splits.numpy(),
                  neighbors_importance=imp, align_corners=align_corners, coordinate_mapping=coordinate_mapping,
                  interpolation=interpolation, normalize=normalize)
        if symmetric:
            k = O.mirror_kernel(k, sym_axis)
            y = O.continuous_conv(k, inp_features=f, **kw)
            g = O.continuous_conv(k.reshape(*k.shape[:3], 1, -1), inp_features=np.ones_like(f[:, :1]), **kw)
            y = y + np.einsum("nc,nco->no", f[:n_out], g.reshape(-1, k.shape[-2], k.shape[-1])).astype(np.float32)
        else:
            y = O.continuous_conv(k, inp_features=f, **kw)
        if bias is not None:
            y = y + bias.numpy()
        y = torch.from_numpy(y.astype(np.float32))
        if out is not None:
            out.copy_(out + y if accumulate else y)
            return out
        return y

    def window_sum(points, queries, radius, window=None, ignore_query_point=False, hash_table=None):
        idx, rs, d = O.fixed_radius_search(points.numpy(), queries.numpy(), radius, ignore_query_point)
        q = d / (np.float32(radius) * np.float32(radius))
        w = np.ones_like(q) if window is None else (d if window == "explicit" else O.window(window, q))
        seg = np.repeat(np.arange(queries.shape[0]), np.diff(rs))
        return torch.from_numpy(np.bincount(seg, weights=w.astype(np.float64), minlength=queries.shape[0]).astype(np.float32))

    def farthest_point_sample(npoint, inp):
        return torch.from_numpy(O.farthest_point_sample(npoint, inp[0].numpy())).unsqueeze(0)

    def gather_point(inp, idx):
        return inp[:, idx[0].long()]

    class GhostSelection:  # ops.ghost_select in torch (what csrc/ghost.hip is tested against on the GPU)
        def __init__(self, pos, boxes, widths2):
            from dmcf_amd import parallel
            if widths2[0] < 0:  # ownership
                inside = ((pos[None] >= boxes[:, None, :3]) & (pos[None] < boxes[:, None, 3:])).all(dim=2)
                self.hits = [torch.nonzero(inside)]
            else:
                gap2 = parallel._gap2_all(pos, torch.stack([boxes[:, :3], boxes[:, 3:]], dim=2))
                self.hits = [torch.nonzero(gap2 <= float(np.float32(v))) for v in widths2]
            self.totals = torch.stack([torch.bincount(h[:, 0], minlength=boxes.shape[0]) for h in self.hits])

        def write(self, sizes):
            assert [int(v) for v in sizes] == [int(h.shape[0]) for h in self.hits]
            return [h[:, 1].contiguous() for h in self.hits]

    monkeypatch.setattr(ops, "ghost_select", lambda pos, boxes, widths2: GhostSelection(pos, boxes.reshape(-1, 6), widths2))
    monkeypatch.setattr(ops, "farthest_point_sample", farthest_point_sample)
    monkeypatch.setattr(ops, "gather_point", gather_point)
    monkeypatch.setattr(ops, "window_sum", window_sum)
    monkeypatch.setattr(ops, "fixed_radius_search", fixed_radius_search)
    monkeypatch.setattr(ops, "build_spatial_hash_table", build_spatial_hash_table)
    monkeypatch.setattr(ops, "cconv_forward", cconv_forward)
    monkeypatch.setattr(ops, "neighbor_counts", lambda r: torch.diff(r.neighbors_row_splits if isinstance(r, ops.NeighborSearchResult) else r).to(torch.float32))
