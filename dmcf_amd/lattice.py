"""Bookkeeping for the lattice form of ContinuousConv (dmcf_lattice_conv_forward, csrc/cconv_lat.hip).

``grid_pos`` (utils/tools/losses.py:136-181) returns ``float(cell) * voxel + center``: a regular lattice.  HRNet convolves
between such point sets (models/hrnet.py:85-92) and, with ``centralize``, all scales of a step share one centre while their
voxel sizes are ``voxel_size * stride`` (losses.py:266) -- integer multiples of each other.  ``ops.grid_pos`` registers what
it knows about its result here; :func:`pair` tells :class:`ContinuousConv` whether two position tensors are such lattices
and hands out the integer cells / the cell -> point table the kernel needs (built once per lattice and step).
"""
import collections
import os
import threading

import numpy as np
import torch

_TLS = threading.local()  # per thread: the virtual ranks of dmcf_amd.parallel are threads, each with its own lattices
_KEEP = 16

# Parity guard.  The reference (and the neighbour-list kernels) subtract two ROUNDED float32 positions,
# fl(fl(n_j v) + c) - fl(fl(n_i v) + c); the lattice form uses the nominal d * v.  The two differ by the rounding of the
# positions, ~ulp(|x|), i.e. ~6e-8 |x| / extent of the filter extent, and the ball -> cube map and the x3 filter scale turn
# that into a relative deviation of the layer output of about 0.6e-6 * |x| / extent (measured on the 1M-particle bench
# scene: 8.5e-6 at |x| / extent = 15.5).  Above this ratio the layer keeps the neighbour-list form,
# whose arithmetic is the reference's own: the deviation of a layer stays below 2e-5 of its output, which is below 1e-7
# of the positions after the out_scale of every shipped model.  DMCF_LATTICE_MAX_RATIO overrides it.
MAX_X_OVER_EXTENT = float(os.environ.get("DMCF_LATTICE_MAX_RATIO", "32"))
# The dense input volume of a launch: at most this many floats (cells x channels; the kernel addresses it with 32-bit byte
# offsets, DMCF_EUNSUPPORTED above 2^29).  A few particles far from the rest -- a splash -- blow the lattices' bounding box up:
# the layer then keeps the neighbour-list form, which does not care.
MAX_VOLUME_FLOATS = 1 << 28
# ... and sparser than one point per MAX_CELLS_PER_POINT cells the zero-filled volume costs more than the list it replaces
MAX_CELLS_PER_POINT = 32
# Stray particles blow a lattice's bounding box up: the dense volume over it grows with the box and every stray lattice point sits
# alone in a 16-cell tile that walks the whole stencil.  Below CORE_MIN_FILL (points per cell of the box) a lattice is split: the
# CORE -- per axis the run of slabs around the fullest one that hold at least CORE_SLAB_FRACTION of its points -- keeps the stencil
# form on a volume that covers the core only, the points outside it (LatticeInfo.core) go through the neighbour-list form, whose
# work follows the neighbours they actually have.  The default threshold is where the form used to give up altogether (one point
# per MAX_CELLS_PER_POINT cells: the dissolved dam break of config 4, 575k lattice points around 148k particles): there the split
# replaces four neighbour-list layers over ALL points.  Between that and a compact scene the split was measured and LOSES
# (round 5, the bench's 1M box while ~15 % of its fluid leaks through the shell, fill 0.3 - 0.6: the four lattice layers go
# 2.6 -> 9.9 ms per step unsplit; split, the lattice launches stay at 2.6 -> 4.4 ms but the stray sets' searches -- a new size
# class every other step, each an exact two-pass search with a host round trip -- and their launches cost more than that:
# 69 - 78 ms per step against 60): DMCF_LATTICE_CORE_FILL=0.6 reproduces it.
CORE_MIN_FILL = float(os.environ.get("DMCF_LATTICE_CORE_FILL", str(1.0 / 32)))
CORE_SLAB_FRACTION = 0.125
_FAR = 3.0e4  # a coordinate no scene point has (the padding rows of an estimated stray set)


def _registry():
    """(data_ptr, n) -> LatticeInfo; the newest few lattices of this thread only."""
    r = getattr(_TLS, "registry", None)
    if r is None:
        r = _TLS.registry = collections.OrderedDict()
    return r


def clear():
    """Forget this thread's lattices (the per-step neighbour cache calls it when a step ends: the positions of the next
    step are new tensors)."""
    _registry().clear()


def _cores():
    """(rollout, voxel) -> dict(box = (lo, hi) cells, cap = stray rows) the PREVIOUS step's histogram chose (LatticeInfo.core)."""
    r = getattr(_TLS, "cores", None)
    if r is None:
        r = _TLS.cores = {}
    return r


def _pick_core(minp, dims, hists):
    """Per axis the run of slabs around the fullest one that hold at least CORE_SLAB_FRACTION of its points (gaps of up to two
    thinner slabs are bridged; a far slab that happens to be well filled -- a second body of fluid -- does NOT stretch the box
    over the empty space between: it is served as strays)."""
    lo, hi = [], []
    for k in range(3):
        h = hists[k]
        m = max(h) if h else 0
        if m <= 0:
            lo.append(minp[k]); hi.append(minp[k] + dims[k])
            continue
        thr = CORE_SLAB_FRACTION * m
        peak = h.index(m)
        a = b = peak
        gap = 0
        i = peak - 1
        while i >= 0 and gap <= 2:
            if h[i] >= thr:
                a, gap = i, 0
            else:
                gap += 1
            i -= 1
        gap = 0
        i = peak + 1
        while i < len(h) and gap <= 2:
            if h[i] >= thr:
                b, gap = i, 0
            else:
                gap += 1
            i += 1
        lo.append(minp[k] + a); hi.append(minp[k] + b + 1)
    return lo, hi


class LatticeInfo:
    def __init__(self, gpos, center, voxel, family, minp, dims, keep=None, center_host=None):
        self.gpos = gpos            # keeps the storage alive, so the registry key stays unique
        # largest |coordinate| any point of the box can have (host arithmetic; None: the centre is not known on the host
        # and the guard of pair() then keeps the neighbour-list form)
        self.max_abs = None
        if center_host is not None:
            self.max_abs = max(abs(float(center_host[k])) + max(abs(int(minp[k])), abs(int(minp[k]) + int(dims[k]))) * float(voxel[k])
                               for k in range(3))
        self.keep = keep            # ... and whatever the family key points at (the source positions / the centre)
        self.version = gpos._version
        self.center = center        # float32 [3] on the device
        self.voxel = tuple(float(np.float32(v)) for v in voxel)
        self.family = family        # lattices of one family share the centre exactly
        self.minp = [int(v) for v in minp]  # (x, y, z) corner of a box of cells that holds every point
        self.dims = [int(v) for v in dims]
        self._cells = None
        self._lin = None
        self._vlin = {}
        self._table = None
        self._core = None
        self._tables = {}

    def cells(self):
        """int32 [n, 3] (x, y, z): the integer lattice coordinates (exact: |gpos - center| / voxel is within 1e-4 of them)."""
        if self._cells is None:
            v = torch.tensor(self.voxel, dtype=torch.float32, device=self.gpos.device)
            self._cells = torch.round((self.gpos - self.center) / v).to(torch.int32).contiguous()
        return self._cells

    def lin(self, minp=None, dims=None):
        """int64 [n]: the points' positions in a dense [dz, dy, dx] array over the box (default: the lattice's own box)."""
        own = minp is None
        if own and self._lin is not None:
            return self._lin
        minp, dims = (self.minp, self.dims) if own else (minp, dims)
        dx, dy, dz = dims
        c = self.cells().long()
        lin = ((c[:, 2] - minp[2]) * dy + (c[:, 1] - minp[1])) * dx + (c[:, 0] - minp[0])
        if os.environ.get("DMCF_LATTICE_CHECK") == "1":  # a point outside the box would scatter out of bounds below
            lo = torch.tensor(list(minp), device=c.device)
            hi = lo + torch.tensor(list(dims), device=c.device)
            if not bool(((c >= lo) & (c < hi)).all()):
                raise RuntimeError("lattice points outside the registered box of cells")
        if own:
            self._lin = lin
        return lin

    def core(self):
        """(min, dims, stray rows int64 [m] | None, stray positions [m, 3] | None, valid float32 [m, 1] | None): the box of cells
        the stencil form serves and the points outside it (None: the whole lattice is the core -- the usual case: a compact
        scene).  Inside a rollout step (estimated sizes, utils/convolutions.neighbor_cache(estimate=True)) the box and the
        capacity m of the stray set are what the PREVIOUS step reported -- any box gives exact results, what lies outside is
        convolved through a neighbour list -- so nothing here waits for the device; rows past the actual count repeat row 0
        with ``valid`` = 0, and a capacity outgrown repeats the step.  Otherwise two host round trips choose both exactly."""
        if self._core is not None:
            return self._core
        n = self.gpos.shape[0]
        dx, dy, dz = self.dims
        whole = (list(self.minp), list(self.dims), None, None, None)
        if n / float(dx * dy * dz) >= CORE_MIN_FILL or os.environ.get("DMCF_LATTICE_CORE", "1") == "0":
            self._core = whole
            return whole
        from .utils.convolutions import _CACHE
        dev = self.gpos.device
        c = self.cells()
        minp, dims = list(self.minp), list(self.dims)
        lo_box = torch.tensor(minp, dtype=torch.int32, device=dev)
        # (a registered box that does not hold every point -- register_points with a caller's box -- must not reach bincount with a
        # negative entry: such points are counted in the box's border cells, i.e. as far out as the histogram can say)
        rel = torch.minimum((c - lo_box).long().clamp_(min=0), torch.tensor(dims, device=dev) - 1)
        hist = torch.cat([torch.bincount(rel[:, k], minlength=dims[k])[: dims[k]] for k in range(3)])
        key = (_CACHE.key, self.voxel)

        def split(h):
            return [h[: dx], h[dx: dx + dy], h[dx + dy:]]

        def next_entry(h):
            """the next step's box and capacity from this step's histogram: strays <= sum over the axes of the points outside the
            box's range on that axis (doubled, + 64k: a spray grows by tens of per cent per step; the padding rows sit far from
            every point and cost a search and a convolution with empty rows)"""
            hs = split(h)
            lo, hi = _pick_core(minp, dims, hs)
            bound = sum(n - sum(hs[k][lo[k] - minp[k]: hi[k] - minp[k]]) for k in range(3))
            return dict(box=(lo, hi), cap=int(2 * bound + 65536))

        est = _cores().get(key) if (_CACHE.depth > 0 and _CACHE.use_hints) else None
        if est is None:
            est = next_entry(hist.tolist())  # (exact: this step's own histogram; one round trip)
            exact = True
        else:
            exact = False
        lo = [max(est["box"][0][k], minp[k]) for k in range(3)]
        hi = [min(est["box"][1][k], minp[k] + dims[k]) for k in range(3)]
        if any(hi[k] <= lo[k] for k in range(3)):
            lo, hi = minp, [minp[k] + dims[k] for k in range(3)]
        lo_t = torch.tensor(lo, dtype=torch.int32, device=dev)
        hi_t = torch.tensor(hi, dtype=torch.int32, device=dev)
        outside = ~((c >= lo_t) & (c < hi_t)).all(dim=1)
        count = outside.sum().reshape(1)
        if exact:
            m = int(count.item())
            _cores()[key] = est  # (serves the next step of this rollout as its estimate)
            valid = None
        else:
            m = min(est["cap"], n)
            cap = m

            def report(vals):
                _cores()[key] = next_entry(vals[1:])
                return vals[0] > cap
            _CACHE.report(torch.cat([count, hist]), report)
        if m == 0:
            idx = pos = valid = None
        else:
            idx = torch.nonzero_static(outside, size=m, fill_value=-1).flatten()
            if not exact:
                valid = (idx >= 0).to(torch.float32).unsqueeze(1)
                idx = idx.clamp(min=0)
            pos = self.gpos.index_select(0, idx).contiguous()
            if valid is not None:  # the padding rows: far from every point (empty rows) -- beyond the lattice's own box
                far = max(_FAR, 4.0 * (self.max_abs or 0.0) + 1.0e3)
                pos = torch.where(valid > 0, pos, torch.full_like(pos, far))
        self._core = (lo, [hi[k] - lo[k] for k in range(3)], idx, pos, valid)
        return self._core

    def table_in(self, minp, dims):
        """int32 [dz, dy, dx] over the given box of cells: the index of the point in each cell, -1 where there is none; points
        outside the box are left out."""
        key = (tuple(minp), tuple(dims))
        t = self._tables.get(key)
        if t is None:
            dx, dy, dz = dims
            c = self.cells().long()
            lo = torch.tensor(list(minp), device=c.device)
            hi = lo + torch.tensor(list(dims), device=c.device)
            ok = ((c >= lo) & (c < hi)).all(dim=1)
            lin = ((c[:, 2] - minp[2]) * dy + (c[:, 1] - minp[1])) * dx + (c[:, 0] - minp[0])
            lin = torch.where(ok, lin, torch.full_like(lin, dz * dy * dx))  # (a spare slot past the end takes the others)
            t = torch.full((dz * dy * dx + 1,), -1, dtype=torch.int32, device=c.device)
            t[lin] = torch.where(ok, torch.arange(c.shape[0], dtype=torch.int32, device=c.device), torch.full((1,), -1, dtype=torch.int32, device=c.device))
            t = self._tables[key] = t[:-1].view(dz, dy, dx)
        return t

    def volume_in(self, features, minp, dims):
        """As :meth:`volume` over a box that need NOT hold every point: the points outside it are left out."""
        key = ("in", tuple(minp), tuple(dims))
        lin = self._vlin.get(key)
        dx, dy, dz = dims
        if lin is None:
            c = self.cells().long()
            lo = torch.tensor(list(minp), device=c.device)
            hi = lo + torch.tensor(list(dims), device=c.device)
            ok = ((c >= lo) & (c < hi)).all(dim=1)
            lin = ((c[:, 2] - minp[2]) * dy + (c[:, 1] - minp[1])) * dx + (c[:, 0] - minp[0])
            lin = self._vlin[key] = torch.where(ok, lin, torch.full_like(lin, dz * dy * dx))
        v = features.new_zeros((dz * dy * dx + 1, features.shape[1]))
        v[lin] = features
        return v[:-1].view(dz, dy, dx, features.shape[1])

    def table(self):
        """int32 [dz, dy, dx]: index of the point in each cell of the box, -1 where there is none."""
        if self._table is None:
            dx, dy, dz = self.dims
            t = torch.full((dz * dy * dx,), -1, dtype=torch.int32, device=self.gpos.device)
            t[self.lin()] = torch.arange(self.gpos.shape[0], dtype=torch.int32, device=self.gpos.device)
            self._table = t.view(dz, dy, dx)
        return self._table

    def volume(self, features, minp=None, dims=None):
        """float32 [dz, dy, dx, C]: ``features`` [n, C] by cell over the box (default: the lattice's own), zeros where there
        is no point.  The box must hold every point."""
        key = None if minp is None else (tuple(minp), tuple(dims))
        lin = self._vlin.get(key)
        if lin is None:
            lin = self._vlin[key] = self.lin(minp, dims)
        dx, dy, dz = self.dims if minp is None else dims
        v = features.new_zeros((dz * dy * dx, features.shape[1]))
        v[lin] = features
        return v.view(dz, dy, dx, features.shape[1])


def register(gpos, center, voxel, family, minp, dims, keep=None, center_host=None):
    if gpos.shape[0] == 0 or any(not (float(v) > 1e-5) for v in voxel):
        return  # empty, or a collapsed axis (2-D scenes): the neighbour-list form handles those
    reg = _registry()
    reg[(gpos.data_ptr(), gpos.shape[0])] = LatticeInfo(gpos, center, voxel, family, minp, dims, keep, center_host)
    while len(reg) > _KEEP:
        reg.popitem(last=False)


def register_points(pos, center, voxel, family, box=None, keep=None, center_host=None):
    """Register ANY float32 [n, 3] tensor of points of the lattice ``center + cell * voxel`` (a filtered grid_pos result,
    owned + ghost points of a sharded step).  ``box`` = (minp, dims) of a box of cells known to hold them all; without it
    the bounding box of the cells is found on the device (one small host round trip)."""
    if pos.shape[0] == 0 or any(not (float(v) > 1e-5) for v in voxel) or not pos.is_cuda:
        return
    if box is None:
        v = torch.tensor([float(np.float32(x)) for x in voxel], dtype=torch.float32, device=pos.device)
        cells = torch.round((pos - center) / v).to(torch.int32)
        lo, hi = cells.amin(dim=0), cells.amax(dim=0)
        b = torch.cat([lo, hi - lo + 1]).tolist()
        box = (b[0:3], b[3:6])
    info = LatticeInfo(pos, center, voxel, family, box[0], box[1], keep if keep is not None else center, center_host)
    reg = _registry()
    reg[(pos.data_ptr(), pos.shape[0])] = info
    while len(reg) > _KEEP:
        reg.popitem(last=False)


def lookup(t):
    info = _registry().get((t.data_ptr(), t.shape[0]))
    if info is None or info.gpos._version != info.version or t.dim() != 2 or t.dtype != torch.float32 or not t.is_contiguous():
        return None
    return info


def enabled():
    """The stencil form takes every lattice point within the radius -- the neighbour set of the default search.  Under an
    emulation of open3d's float walk (ops.search_set() != "distance") the layers keep their neighbour lists, so that a step
    follows ONE reading of the reference end to end."""
    from . import ops
    return os.environ.get("DMCF_LATTICE_CONV", "1") != "0" and ops.search_set() == "distance"


class LatticePair:
    """``inp`` -> ``out`` with ``ratio`` = output spacing / input spacing: 1, 2, ... (outputs on the same or a coarser
    lattice) or 0.5 (outputs on the 2x finer lattice)."""

    def __init__(self, inp, out, ratio):
        self.inp, self.out, self.ratio = inp, out, ratio

    def plan(self, ops, extent, dev):
        """(volume min, volume dims, parts | None) of the launch for this extent: the box of input cells the kernel can touch
        (outputs on the finer lattice: the eight parity classes of the output cells, each with its own stencil, one volume
        that serves them all)."""
        key = (float(extent), str(dev))
        hit = getattr(self, "_plan", None)
        if hit is not None and hit[0] == key:
            return hit[1]
        a, b = self.inp, self.out
        radius = 0.5 * float(extent)
        omin, odim = b.core()[:2]  # the output cells this form serves (the whole lattice unless strays blew its box up)
        cropped = self.cropped = (omin, odim) != (list(b.minp), list(b.dims)) or a.core()[2] is not None
        if self.ratio >= 1:
            step = int(self.ratio)
            if cropped:  # the volume covers what the core's stencils reach, whatever lies outside is left out of it
                vmin, vdim = ops.lattice_volume_box(omin, odim, step, ops.lattice_reach(a.voxel, radius, dev))
            else:
                vmin, vdim = ops.lattice_volume_box(b.minp, b.dims, step, ops.lattice_reach(a.voxel, radius, dev), a.minp, a.dims)
            res = (vmin, vdim, None)
        else:
            lo = [omin[k] for k in range(3)]
            hi = [omin[k] + odim[k] - 1 for k in range(3)]
            if cropped:
                launches, vlo, vhi = [], [1 << 30] * 3, [-(1 << 30)] * 3
            else:
                launches, vlo, vhi = [], list(a.minp), [a.minp[k] + a.dims[k] - 1 for k in range(3)]
            for pz in (0, 1):
                for py in (0, 1):
                    for px in (0, 1):
                        ph = (px, py, pz)
                        # base vectors a with lo <= 2 a + phase <= hi
                        bmin = [-((ph[k] - lo[k]) // 2) for k in range(3)]
                        bmax = [(hi[k] - ph[k]) // 2 for k in range(3)]
                        bdim = [bmax[k] - bmin[k] + 1 for k in range(3)]
                        if min(bdim) <= 0:
                            continue
                        shift = [ph[k] * b.voxel[k] for k in range(3)]
                        m, d = ops.lattice_volume_box(bmin, bdim, 1, ops.lattice_reach(a.voxel, radius, dev, shift))
                        vlo = [min(vlo[k], m[k]) for k in range(3)]
                        vhi = [max(vhi[k], m[k] + d[k] - 1) for k in range(3)]
                        launches.append((ph, bmin, bdim, shift))
            vdim = [vhi[k] - vlo[k] + 1 for k in range(3)]
            parts = [dict(out_phase=ph, rel_shift=shift, base_min=bmin, base_dims=bdim) for ph, bmin, bdim, shift in launches]
            res = (vlo, vdim, parts)
        self._plan = (key, res)
        return res

    def volume_cells(self, ops, extent, dev):
        vdim = self.plan(ops, extent, dev)[1]
        return int(vdim[0]) * int(vdim[1]) * int(vdim[2])

    def conv(self, ops, kernel, inp_features, n_out, extent, **kw):
        """The launch of dmcf_lattice_conv_forward for this pair (outputs on the finer lattice: one
        dmcf_lattice_conv_forward_batch grid of the eight parity classes)."""
        a, b = self.inp, self.out
        vmin, vdim, parts = self.plan(ops, extent, inp_features.device)
        if self.cropped:
            omin, odim = b.core()[:2]
            vol, table, tmin = a.volume_in(inp_features, vmin, vdim), b.table_in(omin, odim), omin
            fill = min(1.0, a.gpos.shape[0] / float(max(vdim[0] * vdim[1] * vdim[2], 1)))
        else:
            vol, table, tmin = a.volume(inp_features, vmin, vdim), b.table(), b.minp
            fill = a.gpos.shape[0] / float(a.dims[0] * a.dims[1] * a.dims[2])
        if parts is None:
            return ops.lattice_conv(kernel, vol, vmin, table, tmin, n_out, a.voxel, extent, inp_step=int(self.ratio),
                                    fill=fill, **kw)
        return ops.lattice_conv(kernel, vol, vmin, table, tmin, n_out, a.voxel, extent, inp_step=1, out_stride=2,
                                parts=parts, fill=fill, **kw)

    def strays(self):
        """(row indices, positions, validity | None) of the output points outside the core (they take the neighbour-list form), or
        None."""
        _, _, idx, pos, valid = self.out.core()
        return None if idx is None else (idx, pos, valid)


def pair(inp_positions, out_positions, extent=None):
    """LatticePair if both tensors are registered lattices of one family and the spacings are in the ratio 1, 2, 3, ...
    (outputs as fine or coarser) or 1/2 (outputs twice as fine); None otherwise (then the neighbour-list form runs).
    ``extent``: the filter extent of the layer, for the parity guard (MAX_X_OVER_EXTENT): None if the scene reaches too far
    from the origin for this extent."""
    if not enabled():
        return None
    a, b = lookup(inp_positions), lookup(out_positions)
    if a is None or b is None or a.family != b.family:
        return None
    if extent is not None:
        if a.max_abs is None or b.max_abs is None or max(a.max_abs, b.max_abs) > MAX_X_OVER_EXTENT * float(extent):
            return None
    if all(a.voxel[k] == 2.0 * b.voxel[k] for k in range(3)):
        return LatticePair(a, b, 0.5)
    step = round(b.voxel[0] / a.voxel[0])
    if step < 1 or any(b.voxel[k] != step * a.voxel[k] for k in range(3)):
        return None
    return LatticePair(a, b, int(step))
