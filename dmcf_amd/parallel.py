"""Spatially sharded rollout: one process per GPU, ghost-particle exchange over ``torch.distributed``.

The reference has no distributed code at all (one process, one visible GPU,
``run_pipeline.py:87-100``); this module is new design work for the multi-GPU row of the scope table
(SURVEY.md section 8e), kept behind the same per-step surface (``step(state) -> state``).

Why per-layer exchange and not one fat halo: a CConv output depends on *features* within R of the point
and features change from layer to layer; the receptive field of the whole Liquid3d net is
0.1 + 3*0.4 + 0.1 = 1.4 length units against blocks of 2.5-5 units, so recomputing the net on a halo that
wide would multiply the work several times.  Instead every layer reads (owned + ghost) inputs and writes
owned outputs; ghost features are refreshed by one all-to-all-v per layer with the ranks whose region is
within R.  Ghost index lists and ghost positions are built once per step per (point set, halo width) and
reused by every layer that shares them.  On an 8-GPU MI355X node every pair of GPUs has its own xGMI link,
so the exchange is single hop; payloads are n_ghost x Cin x 4 B (MBs), i.e. latency rather than bandwidth
bound, and tiny global reductions (fluid bounding box for the boundary crop, ``pbf_model.py:330-334``; the
lattice origin, ``losses.py:137-139``) are 4-6 float all-reduces.

Correctness by construction:
  * ownership is a pure function of position (slab index along one axis), for particles and for the
    coarse lattice points alike, so every point has exactly one owner;
  * a rank's (owned + ghost) input set contains every point within the layer's radius of any of its owned
    output points (halo width = radius + slack; extra ghosts are harmless, the search tests distances
    exactly), hence every owned output sees exactly the neighbour set it would see on one GPU;
  * ghost copies are bit-identical to the owner's values (they are copies), so the antisymmetric ASCC pair
    terms computed on two ranks cancel exactly as they do on one GPU;
  * results differ from the single-GPU path only through summation order (neighbour order inside a row)
    and the last-ulp rounding of the lattice origin (a distributed float64 sum instead of a float32 mean).

``LocalComm`` runs N virtual ranks as threads of one process on one device: it exists so the sharded path
(decomposition, exchange plans, the real HIP kernels) can be tested against the unsharded result on a
single-GPU box; ``TorchDistComm`` is the production communicator (backend ``nccl`` = RCCL on GPUs, ``gloo``
in the CPU tests).
"""
import threading

import math
import os

import numpy as np
import torch

from .utils.convolutions import neighbor_cache
from . import ops
from . import lattice
from .utils.tools.losses import grid_pos


# --------------------------------------------------------------------------------------------------
# communicators
# --------------------------------------------------------------------------------------------------
class Comm:
    rank = 0
    world = 1

    def all_reduce(self, t, op="sum"):
        """In place all-reduce of a small tensor; op in {'sum', 'min', 'max'}."""
        return t

    def all_to_all(self, send, recv_counts=None):
        """send[r]: tensor for rank r (same dtype / trailing dims everywhere, variable first dim).
        Returns recv with recv[r] = what rank r sent to this rank."""
        return [send[0]]


class TorchDistComm(Comm):
    """torch.distributed communicator: counts then payload, each one all_to_all_single."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def all_reduce(self, t, op="sum"):
        ops = {"sum": self.dist.ReduceOp.SUM, "min": self.dist.ReduceOp.MIN, "max": self.dist.ReduceOp.MAX}
        self.dist.all_reduce(t, op=ops[op], group=self.group)
        return t

    def all_to_all(self, send, recv_counts=None):
        """``recv_counts``: rows each rank will send here, when the caller already knows them (a ghost plan exchanges
        the same rows for every layer): skips the count exchange and its host round trip."""
        dist = self.dist
        dev = send[0].device
        trailing = tuple(send[0].shape[1:])
        width = int(np.prod(trailing)) if trailing else 1
        sc = [int(s.shape[0]) for s in send]
        if recv_counts is None:
            send_counts = torch.tensor(sc, dtype=torch.int64, device=dev)
            rc_t = torch.empty_like(send_counts)
            dist.all_to_all_single(rc_t, send_counts, group=self.group)
            rc = rc_t.tolist()
        else:
            rc = [int(c) for c in recv_counts]
        inp = torch.cat([s.reshape(s.shape[0], width) for s in send], dim=0).contiguous()
        out = torch.empty((sum(rc), width), dtype=inp.dtype, device=dev)
        dist.all_to_all_single(out, inp, output_split_sizes=rc, input_split_sizes=sc, group=self.group)
        parts = torch.split(out, rc, dim=0)
        return [p.reshape((p.shape[0],) + trailing) for p in parts]


class _LocalHub:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world


class LocalComm(Comm):
    """N virtual ranks = N threads of one process sharing one device (test vehicle, see module docstring)."""

    def __init__(self, hub, rank):
        self.hub = hub
        self.rank = rank
        self.world = hub.world

    def all_reduce(self, t, op="sum"):
        hub = self.hub
        hub.slots[self.rank] = t.clone()
        hub.barrier.wait()
        stack = torch.stack([hub.slots[r] for r in range(self.world)])
        res = {"sum": stack.sum(0), "min": stack.min(0).values, "max": stack.max(0).values}[op]
        hub.barrier.wait()
        t.copy_(res)
        return t

    def all_to_all(self, send, recv_counts=None):
        hub = self.hub
        hub.slots[self.rank] = send
        hub.barrier.wait()
        recv = [hub.slots[r][self.rank].clone() for r in range(self.world)]
        hub.barrier.wait()
        return recv


def run_local_ranks(world, fn):
    """Run ``fn(comm)`` on ``world`` virtual ranks (threads); returns the list of results."""
    hub = _LocalHub(world)
    results, errors = [None] * world, [None] * world

    def work(r):
        try:
            results[r] = fn(LocalComm(hub, r))
        except BaseException as e:  # noqa: BLE001
            errors[r] = e
            hub.barrier.abort()

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errors:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in errors:
        if e is not None:
            raise e
    return results


# --------------------------------------------------------------------------------------------------
# decomposition and ghost plans
# --------------------------------------------------------------------------------------------------
class SlabDecomposition:
    """Region of rank r = { x : cuts[r] <= x[axis] < cuts[r+1] }, cuts[0] = -inf, cuts[world] = +inf."""

    def __init__(self, axis, inner_cuts):
        self.axis = int(axis)
        self.inner = [float(c) for c in inner_cuts]
        assert all(a < b for a, b in zip(self.inner, self.inner[1:])), "cuts must increase"
        self.world = len(self.inner) + 1

    @staticmethod
    def uniform(axis, lo, hi, world):
        return SlabDecomposition(axis, [lo + (hi - lo) * r / world for r in range(1, world)])

    def bounds(self, r):
        lo = -float("inf") if r == 0 else self.inner[r - 1]
        hi = float("inf") if r == self.world - 1 else self.inner[r]
        return lo, hi

    def owner(self, pos):
        if self.world == 1:
            return torch.zeros(pos.shape[0], dtype=torch.int64, device=pos.device)
        b = torch.tensor(self.inner, dtype=pos.dtype, device=pos.device)
        return torch.bucketize(pos[:, self.axis].contiguous(), b, right=True)

    def within(self, pos, r, width):
        """mask of points whose distance to the region of rank r is <= width"""
        lo, hi = self.bounds(r)
        x = pos[:, self.axis]
        return (x >= lo - width) & (x < hi + width)


class GhostPlan:
    """Ghost copies of one owned point set for one halo width: built once, reused by every layer."""

    def __init__(self, comm, decomp, pos_owned, width):
        self.comm = comm
        self.decomp = decomp
        width = float(width) * (1.0 + 1e-5) + 1e-6  # superset slack; the search re-tests distances exactly
        self.width = width
        self._in_wide = {}
        empty = torch.zeros(0, dtype=torch.int64, device=pos_owned.device)
        self.send_idx = [empty if r == comm.rank else torch.nonzero(decomp.within(pos_owned, r, width)).reshape(-1)
                         for r in range(comm.world)]
        recv = comm.all_to_all([pos_owned[i] for i in self.send_idx])
        self.recv_counts = [int(r.shape[0]) for r in recv]  # the same rows travel for every layer of the step
        self.n_owned = pos_owned.shape[0]
        self.ghost_pos = torch.cat(recv, dim=0) if comm.world > 1 else pos_owned[:0]
        self.pos_ext = torch.cat([pos_owned, self.ghost_pos], dim=0).contiguous()

    def extend(self, feats_owned):
        """[n_owned, C] -> [n_owned + n_ghost, C] (owned rows first, ghosts in the order of ``pos_ext``)."""
        if self.comm.world == 1:
            return feats_owned
        recv = self.comm.all_to_all([feats_owned[i] for i in self.send_idx], recv_counts=self.recv_counts)
        return torch.cat([feats_owned] + recv, dim=0).contiguous()

    def index_in(self, wide):
        """Rows of ``wide``'s ghosts (a plan of the same point set with a larger width) that are this plan's ghosts, in this
        plan's order: the senders picked their rows with ``decomp.within(pos, this rank, width)`` in ascending row order, and
        the same test on the same received positions picks the same rows here -- no communication."""
        idx = self._in_wide.get(id(wide))
        if idx is None:
            parts, off = [], 0
            for cnt in wide.recv_counts:
                g = wide.ghost_pos[off:off + cnt]
                parts.append(torch.nonzero(self.decomp.within(g, self.comm.rank, self.width)).reshape(-1) + off)
                off += cnt
            idx = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64, device=wide.ghost_pos.device)
            if idx.shape[0] != self.ghost_pos.shape[0] or (os.environ.get("DMCF_SHARD_CHECK") == "1"
                                                           and not torch.equal(wide.ghost_pos[idx], self.ghost_pos)):
                raise RuntimeError("ghost plans of one point set are not nested")
            self._in_wide[id(wide)] = idx
            self._wide_keep = wide  # id() stays unique while the plan lives
        return idx

    def extend_from(self, wide, wide_ext):
        """``extend`` without communication, from the same features already extended by the wider plan."""
        return torch.cat([wide_ext[:self.n_owned], wide_ext[self.n_owned:][self.index_in(wide)]], dim=0)


# --------------------------------------------------------------------------------------------------
# sharded per-step driver for the PBFNet family (SymNet / HRNet / CConv)
# --------------------------------------------------------------------------------------------------
class ShardedSimulator:
    """``step(state) -> state`` on the particles this rank owns.

    state = dict(pos, vel, acc|None, box, box_normals, gid): owned fluid particles (``gid`` = global particle
    id, int64, carried through migration), owned boundary particles (static).  The arithmetic per layer is the
    model's own layers (same weights, same HIP kernels); only *which rows* a rank holds differs."""

    def __init__(self, model, comm, decomp):
        self.model = model
        self.comm = comm
        self.decomp = decomp
        assert decomp.world == comm.world
        self.exchanged_rows = 0

    # -- helpers -----------------------------------------------------------------------------------
    def _plan(self, name, width):
        key = (name, round(float(width), 9))
        plan = self._plans.get(key)
        if plan is None:
            plan = GhostPlan(self.comm, self.decomp, self._sets[name], width)
            self._plans[key] = plan
            lat = self._lattices.get(name)
            if lat is not None:
                box = lat[2]  # the union of all ranks' boxes: owned + ghost points
                lattice.register_points(plan.pos_ext, lat[0], lat[1], ("sharded", lat[0].data_ptr()), box)
        return plan

    def _conv(self, layer, feats_owned, inp, out, extent, share=None):
        """layer(feats, pos[inp] -> pos[out]) with inputs extended by the ghosts within extent/2.
        ``share`` = dict(width=...) common to all the layers that read the SAME features: the ghost rows travel once, at the
        largest width any of them needs, and the narrower sets are subsets of those rows (GhostPlan.index_in)."""
        plan = self._plan(inp, 0.5 * float(extent))
        if share is None or self.comm.world == 1:
            feats_ext = plan.extend(feats_owned)
            self.exchanged_rows += feats_ext.shape[0] - feats_owned.shape[0]
        else:
            wide = self._plan(inp, share["width"])
            if "ext" not in share:
                share["ext"] = wide.extend(feats_owned)
                self.exchanged_rows += share["ext"].shape[0] - feats_owned.shape[0]
            feats_ext = share["ext"] if plan is wide else plan.extend_from(wide, share["ext"])
        return layer(feats_ext, plan.pos_ext, self._sets[out], extent, None)

    def _global_sum(self, t64):
        return self.comm.all_reduce(t64, "sum")

    # -- one step ----------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, state):
        """One time step.  Like Simulator.run_inference the searches run with row capacities estimated from the previous
        step (single pass, no host round trips); whether any rank outgrew an estimate is agreed with ONE tiny all-reduce
        after the step -- the validation happens when the cache scope closes, i.e. after every collective of the step --
        and then all ranks repeat the step with the exact search."""
        try:
            with neighbor_cache(estimate=True):
                out = self._step(state)
            ok = 1
        except ops.NeighborCapacityExceeded:
            out, ok = None, 0
        flag = torch.tensor([ok], dtype=torch.int32, device=state["pos"].device)
        self.comm.all_reduce(flag, "min")
        if int(flag.item()) == 0:
            with neighbor_cache(estimate=False):
                out = self._step(state)
        return out

    def _step(self, state):
        m, comm = self.model, self.comm
        dev = state["pos"].device
        self._plans, self._sets, self._lattices = {}, {}, {}
        pos0, vel0, acc = state["pos"], state["vel"], state.get("acc")
        box_all, bfeats_all = state["box"], state["box_normals"]
        if "grav_eqvar" in m.transformation:
            raise NotImplementedError("grav_eqvar (WBC-SPH) is not wired into the sharded path; 2-D scenes of a few "
                                      "thousand particles do not amortise a halo (SURVEY.md section 8e, last row)")
        d = m.transform([pos0, vel0, acc, None, box_all, bfeats_all])
        _pos, _vel, acc_t, _, box, bfeats = d
        pos, vel = m.integrate_pos_vel(_pos, _vel, acc_t)
        # fluid bounding box over all ranks (pbf_model.py:330-334)
        filter_extent = [float(np.float32(r) * np.float32(2)) for r in m.particle_radii]
        big = 3.0e38
        pt = pos.t().contiguous()  # [3, N]: row reductions (a strided column reduction of [N, 3] is ~0.5 ms each)
        lo = pt.amin(dim=1) if pos.shape[0] else torch.full((3,), big, device=dev)
        hi = pt.amax(dim=1) if pos.shape[0] else torch.full((3,), -big, device=dev)
        lo = comm.all_reduce(lo.clone(), "min") - filter_extent[-1]
        hi = comm.all_reduce(hi.clone(), "max") + filter_extent[-1]
        keep = ((box >= lo) & (box <= hi)).all(dim=1)
        box, bfeats = box[keep], bfeats[keep]

        fluid_feats = [torch.ones_like(pos[:, :1])]
        if m.use_vel:
            fluid_feats.append(vel)
        if m.use_acc:
            fluid_feats.append(acc_t)
        box_feats = [torch.ones_like(box[:, :1])]
        if m.use_box_feats:
            box_feats.append(bfeats)
        fluid_feats = torch.cat(fluid_feats, dim=-1)
        box_feats = torch.cat(box_feats, dim=-1)
        all_pos = torch.cat([pos, box], dim=0).contiguous()
        self._sets.update(pos=pos.contiguous(), box=box.contiguous(), s0=all_pos)
        n_fluid = pos.shape[0]

        operands = m.fused_input_operands(fluid_feats, box_feats)
        if operands is not None:
            # the two input layers as one block-diagonal convolution over all particles (models/pbf_model.py): one ghost
            # plan and one exchange instead of two of each
            in_feats, in_kernel, in_bias = operands
            fused = self._conv(lambda f, pi, po, ext, _: m.fused_input_conv(in_kernel, in_bias, f, pi, po, ext)[0],
                               in_feats, "s0", "s0", filter_extent[0])
            co = m.fluid_convs.filters
            ans_conv, ans_obs = fused[:, :co].contiguous(), fused[:, co:].contiguous()
        else:
            ans_conv = self._conv(m.fluid_convs, fluid_feats * m.part_scale, "pos", "s0", filter_extent[0])
            ans_obs = self._conv(m.obs_convs, box_feats * m.part_scale, "box", "s0", filter_extent[0])
        ans_dense = m.fluid_dense(fluid_feats)
        ans_dense = torch.cat([ans_dense, m.obs_dense(box_feats)], dim=0)
        feats = torch.cat([ans_conv, ans_obs, ans_dense], dim=-1)

        # multi-scale point sets: the same lattice on every rank (global origin), each rank keeps its region
        base = "s0" if m.use_bnds else "pos"
        names = []
        center = None
        if m.centralize and any(s != 1 for s in m.strides):
            acc64 = torch.cat([self._sets[base].t().contiguous().double().sum(dim=1),
                               torch.tensor([float(self._sets[base].shape[0])], dtype=torch.float64, device=dev)])
            acc64 = self._global_sum(acc64)
            center = (acc64[:3] / acc64[3]).to(torch.float32)
        for si, stride in enumerate(m.strides):
            if stride == 1:
                names.append(base)
                continue
            if m.voxel_size is None:
                raise NotImplementedError("FPS based multi-scale (voxel_size None) is out of scope")
            vs = np.asarray(m.voxel_size, dtype=np.float32) * np.float32(stride)
            margin = float(vs.max()) * (1.0 + m.sample_hyst + m.sample_pad + 0.05)
            cand = self._plan(base, margin).pos_ext
            g = grid_pos(cand, vs, centralize=m.centralize, pad=m.sample_pad, hyst=m.sample_hyst, center=center)
            g = g[self.decomp.owner(g) == comm.rank].contiguous()
            name = f"s{si}"
            self._sets[name] = g
            names.append(name)
            if center is not None and g.is_cuda:
                # all lattices of the step share the agreed centre: the layers between them (and their owned + ghost
                # inputs, see _plan) can take the lattice form of ContinuousConv (dmcf_amd/lattice.py)
                box = ops.grid_pos_last_box()  # of the candidates' lattice: holds the owned points
                # the ghost copies a layer adds to this set come from other ranks' lattices: the union of all ranks' boxes
                # (one tiny all-reduce, here where the queue is empty anyway) holds owned and ghost points alike.  Only the
                # INPUT volumes are that large (zero-filled, 4 B x Cin per cell); the kernel walks the output box.
                lo, dims = list(box[0]), list(box[1])
                ext = torch.tensor([v for k in range(3) for v in (-lo[k], lo[k] + dims[k] - 1)], dtype=torch.int64, device=dev)
                ext = comm.all_reduce(ext, "max").tolist()
                for k in range(3):
                    lo[k], dims[k] = -ext[2 * k], ext[2 * k + 1] + ext[2 * k] + 1
                self._lattices[name] = (center, [float(v) for v in vs], (lo, dims))
                lattice.register_points(g, center, vs, ("sharded", center.data_ptr()), (lo, dims))

        out = self._forward(names, feats, filter_extent, n_fluid)

        # postprocess (pbf_model.py:440-489) on the owned fluid particles
        if out.shape[-1] == 1:
            out = out.repeat(1, 3)
        elif out.shape[-1] == 2:
            out = torch.cat([out, out[:, :1]], dim=-1)
        out_scale = torch.tensor(m.out_scale, dtype=torch.float32, device=dev)
        pos_correction = out_scale * out[:n_fluid]
        self.net_output, self.pos_correction = out, pos_correction
        pos2, vel2 = m.integrate_pos_vel(_pos, _vel, acc_t)
        new_pos, new_vel = m.compute_new_pos_vel(_pos, _vel, pos2, vel2, pos_correction)
        new_pos, new_vel = m.inv_transform([new_pos, new_vel], None)

        # migration: every particle goes to the owner of its new position
        own = self.decomp.owner(new_pos)
        gid = state["gid"]
        payload = torch.cat([new_pos, new_vel] + ([acc] if acc is not None else []), dim=1)
        send_idx = [torch.nonzero(own == r).reshape(-1) for r in range(comm.world)]
        recv = comm.all_to_all([payload[i] for i in send_idx])
        recv_gid = comm.all_to_all([gid[i] for i in send_idx])
        payload = torch.cat(recv, dim=0)
        new_state = dict(pos=payload[:, 0:3].contiguous(), vel=payload[:, 3:6].contiguous(),
                         acc=payload[:, 6:9].contiguous() if acc is not None else None,
                         box=box_all, box_normals=bfeats_all, gid=torch.cat(recv_gid, dim=0))
        return new_state

    def _forward(self, names, feats, filter_extent, n_fluid):
        m = self.model
        kind = type(m).__name__
        if kind in ("SymNet", "HRNet"):
            if not m.use_bnds:
                feats = feats[:n_fluid]
            ans_convs = [[feats]]
            for layer in range(len(m.convs)):
                ans = []
                relu_in = [torch.relu(t) for t in ans_convs[-1]]  # once per layer (see models/hrnet.py)
                # every output scale reads the same relu(x_{inp_scale}): one ghost exchange per input scale and layer, at
                # the largest radius of the layer, instead of one per (scale, inp_scale)
                n_scales = len(m.convs[layer])
                shares = [dict(width=0.5 * float(max(filter_extent[max(i, s)] for s in range(n_scales))))
                          for i in range(len(relu_in))]
                for scale in range(n_scales):
                    if len(m.convs[layer][scale]) != 1:
                        raise NotImplementedError("k > 0 sub-layers (hrnet.py:120-131) are unused by shipped configs")
                    importance = m.part_scale if scale == 0 else 1.0
                    inp = []
                    for inp_scale in range(len(ans_convs[-1])):
                        f = relu_in[inp_scale]
                        ext = filter_extent[max(inp_scale, scale)]
                        conv_in = f if importance == 1.0 else f * importance
                        ans_conv = self._conv(m.convs[layer][scale][0][inp_scale], conv_in, names[inp_scale],
                                              names[scale], ext, share=shares[inp_scale] if conv_in is f else None)
                        if scale == inp_scale:
                            ans_conv = ans_conv + m.denses[layer][scale][0][inp_scale](f)
                            if ans_conv.shape[-1] == ans_convs[-1][scale].shape[-1]:
                                ans_conv = ans_conv + ans_convs[-1][scale]
                        inp.append(ans_conv)
                    if m.add_merge:
                        merged = inp[0]
                        for t in inp[1:]:
                            merged = merged + t
                        ans.append(merged)
                    else:
                        ans.append(torch.cat(inp, dim=-1))
                ans_convs.append(ans)
            out = m.out_activation(ans_convs[-1][0])
            if kind == "SymNet":
                if not m.use_bnds:
                    raise NotImplementedError("use_bnds=False with the ASCC head in the sharded path")
                ext = float(np.float32(m.particle_radii[0]) * np.float32(2))
                for conv in m.sym_convs:
                    out = torch.relu(out)
                    conv_in = out if m.part_scale == 1.0 else out * m.part_scale
                    out = self._conv(conv, conv_in, "s0", "s0", ext)
                out = m.act(out)
            return out
        if kind == "CConv":
            feats = feats[:n_fluid]
            ext = float(np.float32(m.particle_radii[0]) * np.float32(2))
            ans_convs = [feats]
            for conv, dense in zip(m.convs, m.denses):
                f = torch.relu(ans_convs[-1])
                ans_conv = self._conv(conv, f, "pos", "pos", ext)
                ans_dense = dense(f)
                ans = ans_conv + ans_dense
                if ans_dense.shape[-1] == ans_convs[-1].shape[-1]:
                    ans = ans + ans_convs[-1]
                ans_convs.append(ans)
            return m.out_activation(ans_convs[-1])
        raise NotImplementedError(kind)


def shard_scene(scene, decomp, rank, device):
    """Owned part of a scene dict(pos, vel, box, box_normals[, acc]) for ``rank`` (numpy in, tensors out)."""
    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
    pos, box = t(scene["pos"]), t(scene["box"])
    own_p = decomp.owner(pos) == rank
    own_b = decomp.owner(box) == rank
    state = dict(pos=pos[own_p].contiguous(), vel=t(scene["vel"])[own_p].contiguous(),
                 acc=t(scene["acc"])[own_p].contiguous() if scene.get("acc") is not None else None,
                 box=box[own_b].contiguous(), box_normals=t(scene["box_normals"])[own_b].contiguous(),
                 gid=torch.nonzero(own_p).reshape(-1))
    return state
