"""Window functions and multi-scale point sets -- mirror of the hot-path part of the reference's
``utils/tools/losses.py`` (same function names and arguments).

  get_window_func   losses.py:8-44     poly6 / cubic / linear / peak / cubic_grad on q = d^2/R^2
  grid_pos          losses.py:136-181  dilated voxel-corner lattice, de-duplicated (tf.unique order)
  get_dilated_pos   losses.py:249-284  one point set per stride (lattice with voxel_size, farthest point sampling without)

  compute_density   losses.py:285-306  windowed neighbour sum (fused into the search scan: ops.window_sum)
  compute_pressure  losses.py:367-377  Tait-style pressure from the density
  density_loss      losses.py:380-398  validation metric of pipelines/simulator.py:227-243

The training losses (losses.py:47-110, 400-414: Chamfer / EMD / get_loss) need the reference's custom CUDA ops
and stay out of scope.
"""
import numpy as np
import torch


class WindowFunction:
    """Callable returned by :func:`get_window_func`.

    Calling it on a tensor of normalised squared distances evaluates the formula with torch (generic
    use, e.g. user code).  ``ContinuousConv`` recognises the object and instead passes ``name`` /
    ``fac`` to the HIP kernel, which evaluates the same formula per neighbour inside the splat
    (dmcf_amd/csrc/cconv.hip: window_value) so no [P]-sized importance array is materialised.
    """

    def __init__(self, name, fac=1.0):
        self.name = name
        self.fac = float(fac)

    def __call__(self, q):
        fac = self.fac
        if self.name == "poly6":  # losses.py:11-12
            return fac * torch.clamp((1 - q) ** 3, 0, 1)
        if self.name == "cubic":  # losses.py:15-20
            s = torch.sqrt(q)
            inner = torch.where(s <= 0.5, 6 * (s ** 3 - q) + 1, 2 * (1 - s) ** 3)
            return fac * 4 / 3 * torch.where(q <= 1, inner, torch.zeros_like(s))
        if self.name == "linear":  # losses.py:23-25
            return fac * (1 - torch.sqrt(q))
        if self.name == "peak":  # losses.py:28-30
            return fac * (1 - 2 * torch.sqrt(q) + q)
        if self.name == "cubic_grad":  # losses.py:33-39
            s = torch.sqrt(q)
            inner = torch.where(s <= 0.5, 18 * q - 12 * s, -6 * (1 - s) ** 2)
            return fac * 4 / 3 * torch.where(q <= 1, inner, torch.zeros_like(s))
        raise NotImplementedError(self.name)

    def __repr__(self):
        return f"WindowFunction({self.name!r}, fac={self.fac})"


def get_window_func(typ, fac=1.0, **kwargs):
    """losses.py:8-44.  ``typ is None`` -> None (no window), unknown -> NotImplementedError."""
    if typ is None:
        return None
    if typ in ("poly6", "cubic", "linear", "peak", "cubic_grad"):
        return WindowFunction(typ, fac)
    raise NotImplementedError()


def _unique_first_occurrence(idx):
    """tf.unique(idx)[0]: unique values in order of first appearance (losses.py:171)."""
    uniq, inverse = torch.unique(idx, sorted=True, return_inverse=True)
    first = torch.full((uniq.shape[0],), idx.shape[0], dtype=torch.int64, device=idx.device)
    first.scatter_reduce_(0, inverse, torch.arange(idx.shape[0], device=idx.device), reduce="amin")
    return uniq[torch.argsort(first)]


def grid_pos(pos, voxel_size, centralize=False, pad=0, hyst=0.1, center=None, return_box=False):
    """losses.py:136-181: lattice corners of the voxels (edge ``voxel_size``) that contain a particle,
    with +-``hyst`` hysteresis; axes with voxel_size < 1e-5 collapse (2-D / 1-D scenes).
    ``center`` (extension used by the sharded path): the lattice origin to use instead of this call's own
    mean when ``centralize`` -- every rank passes the global mean so all ranks build the same lattice.
    ``return_box``: -> (points, (minp, dims) | None), the integer box of cells (x, y, z) that holds every returned point;
    None when it is not known on the host (empty result, or the sort-based fallback ran)."""
    # voxel_size is kept on the host (list / numpy float32) so that the axis collapse test does not
    # force a device round trip; values are float32 like the reference's tf.constant
    vs_host = np.asarray(voxel_size.detach().cpu() if isinstance(voxel_size, torch.Tensor) else voxel_size,
                         dtype=np.float32).reshape(3)
    if pos.is_cuda:  # product path: the HIP kernels (csrc/grid.hip); below is the host-side form of the same
        from ... import ops
        try:
            out, box = ops.grid_pos(pos, vs_host, centralize=centralize, pad=pad, hyst=hyst, center=center, return_box=True)
            return (out, box) if return_box else out
        except ops.GridTooSparse:
            pass  # bounding box too large for a dense cell table: sort-based form, still on the device
    voxel_size = torch.from_numpy(vs_host.copy()).to(pos.device)
    if centralize:
        if center is None:
            center = pos.mean(dim=0)  # :138
        pos = pos - center
    active = voxel_size >= 1e-5
    vs = torch.clamp(voxel_size, min=1e-5)
    h = torch.where(active, torch.full_like(vs, hyst), torch.zeros_like(vs))
    scaled = pos / vs
    dpos = torch.cat([torch.floor(scaled - h).to(torch.int32), torch.floor(scaled + h).to(torch.int32)], dim=0)  # :142-150
    active_host = (vs_host >= 1e-5).tolist()
    ranges_host = [list(range(-pad, 2 + pad)) if a else [0] for a in active_host]  # :151-161
    ranges = [torch.tensor(r, device=pos.device) for r in ranges_host]
    offset = torch.stack(torch.meshgrid(*ranges, indexing="ij"), dim=-1).reshape(1, -1, 3).to(torch.int32)
    dpos = (dpos.unsqueeze(1) + offset).reshape(-1, 3)
    # :167-170 -- floor is monotone, so the extrema of the 16N candidate cells follow from the extrema of the N
    # scaled positions (a [16N,3] column reduction is ~8 ms per call on 18M rows; this one is ~0.5 ms)
    smin, smax = scaled.min(dim=0).values, scaled.max(dim=0).values
    off_lo = torch.tensor([r[0] for r in ranges_host], dtype=torch.int32, device=pos.device)
    off_hi = torch.tensor([r[-1] for r in ranges_host], dtype=torch.int32, device=pos.device)
    minp = torch.floor(smin - h).to(torch.int32) + off_lo
    maxp = torch.floor(smax + h).to(torch.int32) + off_hi - minp + 1
    maxp64 = maxp.to(torch.int64)
    mult = torch.stack([torch.ones_like(maxp64[0]), maxp64[0], maxp64[0] * maxp64[1]])
    idx = ((dpos - minp).to(torch.int64) * mult).sum(dim=-1)
    idx = _unique_first_occurrence(idx)  # :171
    gpos = torch.stack([idx % maxp64[0], idx // maxp64[0] % maxp64[1], idx // (maxp64[0] * maxp64[1])], dim=-1) \
        + minp.to(torch.int64)  # :172-174
    if centralize:
        out = gpos.to(torch.float32) * voxel_size + center  # :177
    else:
        out = gpos.to(torch.float32) * voxel_size + voxel_size / 2  # :179
    return (out, None) if return_box else out


def get_dilated_pos(pos, strides, voxel_size=None, centralize=False, pad=0, hyst=0.1):
    """losses.py:249-284 -> (dilated_pos, pcnt, idx).  With ``voxel_size`` every coarse level is a lattice computed from
    the full-resolution set (:266-272; no entry is appended to ``idx`` on that branch, as in the reference); without it
    level k is ``n // stride`` farthest-point samples of level k-1 (:274-282) and ``idx[k]`` the [1, m] sample indices
    HRNet's cross-scale Dense branch uses."""
    from ... import ops
    pcnt, dilated_pos, idx = [], [], []
    lattices = {}
    if voxel_size is not None:
        vs = voxel_size.detach().cpu().numpy() if isinstance(voxel_size, torch.Tensor) else voxel_size
        coarse = [s for s in strides if s != 1]
        if pos.is_cuda and len(coarse) > 1:
            # every coarse level is built from the SAME positions: their kernels are enqueued together and the host waits twice
            # for all of them, not twice per level (a step of the 2-D models is paced by its host round trips)
            try:
                res = ops.grid_pos_many(pos, [np.asarray(vs, dtype=np.float32) * np.float32(s) for s in coarse],
                                        centralize=centralize, pad=pad, hyst=hyst)
                lattices = {s: r[0] for s, r in zip(coarse, res)}
            except ops.GridTooSparse:
                lattices = {}  # (a level's box is too sparse for the dense cell table: one at a time, each in the form it needs)
    for stride in strides:
        if stride == 1:
            pcnt.append(pos.shape[0])
            dilated_pos.append(pos)
            idx.append(None)
        elif voxel_size is not None:
            if stride in lattices:
                dilated_pos.append(lattices[stride])
            else:
                v_scale = np.asarray(vs, dtype=np.float32) * np.float32(stride)  # :266
                dilated_pos.append(grid_pos(pos, v_scale, centralize=centralize, pad=pad, hyst=hyst))
            pcnt.append(dilated_pos[-1].shape[0])
        else:
            sample_cnt = max(pos.shape[0] // stride, 1)  # :275
            pcnt.append(sample_cnt)
            idx.append(ops.farthest_point_sample(sample_cnt, dilated_pos[-1].unsqueeze(0)))
            dilated_pos.append(ops.gather_point(dilated_pos[-1].unsqueeze(0), idx[-1])[0])
    return dilated_pos, pcnt, idx


def compute_density(out_pos, in_pos=None, radius=0.005, win=None):
    """losses.py:285-306: ``dens[i] = sum_j win(|in_pos[j] - out_pos[i]|^2 / radius^2)`` over the points within
    ``radius`` (the query point itself included).  ``win``: an object from :func:`get_window_func` (evaluated inside
    the search kernel, no pair list), None (the reference warns and uses the identity: sum of q), or any callable
    on a tensor of q values (evaluated with torch on the pair list)."""
    from ... import ops
    if in_pos is None:
        in_pos = out_pos
    radius = float(radius)
    if win is None:
        return ops.window_sum(in_pos, out_pos, radius, "explicit") / (radius * radius)
    if isinstance(win, WindowFunction):
        d = ops.window_sum(in_pos, out_pos, radius, win.name)
        return d if win.fac == 1.0 else d * win.fac
    nns = ops.fixed_radius_search(in_pos, out_pos, radius, return_distances=True)
    return ops.reduce_subarrays_sum(win(nns.neighbors_distance / (radius * radius)), nns.neighbors_row_splits)


def compute_transformed_dx(pos, scale=None, rot=None, radius=0.005):
    """losses.py:337-364: per point the MEAN over its neighbours within ``radius`` (itself included) of ``(x_j - x_i) *
    scale_j``.  The reference's caller passes ``rot=None`` (models/pbf_model.py:458-460 has the quaternion branch commented
    out); a rotation raises here.  The pair list comes from the fixed-radius search, the ragged sums from
    dmcf_reduce_subarrays_sum; a point without neighbours gives 0 / 0 = NaN as tf.reduce_mean over an empty row does."""
    from ... import ops
    if rot is not None:
        raise NotImplementedError("compute_transformed_dx with a rotation (quat_mean / quat_rot): the reference never passes one")
    nns = ops.fixed_radius_search(pos, pos, float(radius), return_distances=False)
    idx, rs = nns.neighbors_index.long(), nns.neighbors_row_splits
    counts = torch.diff(rs)
    row = torch.repeat_interleave(torch.arange(pos.shape[0], device=pos.device), counts, output_size=idx.shape[0])
    dx = pos[idx] - pos[row]
    if scale is not None:
        dx = dx * scale[idx]
    sums = torch.stack([ops.reduce_subarrays_sum(dx[:, k].contiguous(), rs) for k in range(dx.shape[1])], dim=1)
    return sums / counts.to(sums.dtype).unsqueeze(1)


def compute_pressure(out_pts, inp_pts=None, dens=None, rest_dens=3.5, stiffness=20.0, win=None):
    """losses.py:367-377 (note: the reference ignores a radius here too: compute_density's default applies)."""
    if inp_pts is None:
        inp_pts = out_pts
    if dens is None:
        dens = compute_density(out_pts, inp_pts, win=win)
    return torch.relu(stiffness * ((dens / rest_dens) ** 7 - 1))


def density_loss(gt, pred, gt_in=None, pred_in=None, radius=0.005, eps=0.01, win=None, use_max=False, **kwargs):
    """losses.py:380-398."""
    pred_dens = compute_density(pred, pred_in, radius, win=win)
    gt_dens = compute_density(gt, gt_in, radius, win=win)
    rest_dens = gt_dens.max()
    if use_max:
        return torch.abs(pred_dens.max() - rest_dens) / rest_dens
    return torch.relu(pred_dens - rest_dens - eps).mean()
