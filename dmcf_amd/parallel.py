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
class Stats:
    """What a sharded step costs besides its kernels (bench.py --gpus N prints it per rank): device -> host reads (each one
    drains the queue), rows sent / received as ghosts and by the migration, and -- with ``profile`` set -- HIP events around
    every wait for an asynchronous exchange (how long the compute stream was held up by it)."""

    def __init__(self):
        self.host_syncs = 0
        self.profile = False
        self.wait_events = []

    def exposed_wait_ms(self):
        ms = sum(a.elapsed_time(b) for a, b in self.wait_events)
        self.wait_events = []
        return ms


STATS = threading.local()


def stats():
    st = getattr(STATS, "v", None)
    if st is None:
        st = STATS.v = Stats()
    return st


def host(x):
    """tensor -> Python list / number: a counted device -> host read"""
    stats().host_syncs += 1
    return x.tolist() if x.dim() else x.item()


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

    def all_to_all_start(self, send, recv_counts):
        """Start the exchange (the receive counts must be known) and return a zero-argument function that waits for it and
        returns what :meth:`all_to_all` returns.  Communicators without asynchronous collectives run it right away."""
        recv = self.all_to_all(send, recv_counts=recv_counts)
        return lambda: recv

    def exchange_counts(self, counts):
        """``counts`` int64 [world, k] on the device, row r = what this rank announces to rank r.  Returns (the same numbers, the rows
        the other ranks announced to this one) as nested host lists -- ONE device -> host read for both: the announcement is a
        device-side collective, nothing has to be on the host before it."""
        h = host(counts)
        return h, h

    def exchange_rows_start(self, rows, send_counts, recv_counts):
        """The exchange of a ghost plan: ``rows`` [sum(send_counts), C] already in peer order (one gather by the plan's
        concatenated send list), both count lists known on the host.  Returns a function that waits and returns the received
        rows in one piece [sum(recv_counts), C] -- no per-peer gathers, splits or concatenations (a step has 18 of these)."""
        wait = self.all_to_all_start(list(torch.split(rows, [int(c) for c in send_counts], dim=0)), recv_counts)
        return lambda: torch.cat(wait(), dim=0)


class TorchDistComm(Comm):
    """torch.distributed communicator: counts then payload, each one all_to_all_single.  Backend "nccl" (= RCCL over xGMI)
    is the production path; with "gloo" (CPU tests, and the two-processes-on-one-GPU test) device tensors are staged
    through the host, because gloo's all-to-all takes CPU tensors only."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.stage = dist.get_backend(group) == "gloo"

    def all_reduce(self, t, op="sum"):
        ops = {"sum": self.dist.ReduceOp.SUM, "min": self.dist.ReduceOp.MIN, "max": self.dist.ReduceOp.MAX}
        if self.stage and t.is_cuda:
            h = t.cpu()
            self.dist.all_reduce(h, op=ops[op], group=self.group)
            t.copy_(h)
            return t
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
        xdev = torch.device("cpu") if self.stage else dev
        if recv_counts is None:
            send_counts = torch.tensor(sc, dtype=torch.int64, device=xdev)
            rc_t = torch.empty_like(send_counts)
            dist.all_to_all_single(rc_t, send_counts, group=self.group)
            rc = host(rc_t)
        else:
            rc = [int(c) for c in recv_counts]
        inp = torch.cat([s.reshape(s.shape[0], width) for s in send], dim=0).contiguous().to(xdev)
        out = torch.empty((sum(rc), width), dtype=inp.dtype, device=xdev)
        dist.all_to_all_single(out, inp, output_split_sizes=rc, input_split_sizes=sc, group=self.group)
        parts = torch.split(out.to(dev), rc, dim=0)
        return [p.reshape((p.shape[0],) + trailing) for p in parts]

    def exchange_counts(self, counts):
        world, k = counts.shape
        x = counts.contiguous().to(torch.device("cpu") if self.stage else counts.device)
        out = torch.empty_like(x)
        self.dist.all_to_all_single(out, x, group=self.group)
        both = host(torch.cat([x.reshape(-1), out.reshape(-1)]))
        n = world * k
        return ([both[r * k:(r + 1) * k] for r in range(world)], [both[n + r * k:n + (r + 1) * k] for r in range(world)])

    def all_to_all_start(self, send, recv_counts):
        """The payload exchange as an asynchronous collective (RCCL runs it on its own stream, next to whatever the compute
        stream is doing); ``wait()`` makes the compute stream wait for it.  Staged (gloo) exchanges run synchronously."""
        if self.stage:
            return super().all_to_all_start(send, recv_counts)
        dist = self.dist
        dev = send[0].device
        trailing = tuple(send[0].shape[1:])
        width = int(np.prod(trailing)) if trailing else 1
        sc = [int(s.shape[0]) for s in send]
        rc = [int(c) for c in recv_counts]
        inp = torch.cat([s.reshape(s.shape[0], width) for s in send], dim=0).contiguous()
        out = torch.empty((sum(rc), width), dtype=inp.dtype, device=dev)
        work = dist.all_to_all_single(out, inp, output_split_sizes=rc, input_split_sizes=sc, group=self.group, async_op=True)

        def wait():
            st = stats()
            if st.profile:  # how long does the compute stream stand still for this exchange?
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                work.wait()
                e1.record()
                st.wait_events.append((e0, e1))
            else:
                work.wait()  # (the current stream waits; ``inp`` / ``out`` stay referenced by this closure until then)
            return [p.reshape((p.shape[0],) + trailing) for p in torch.split(out, rc, dim=0)]
        return wait


def _torchdist_exchange_rows_start(self, rows, send_counts, recv_counts):
    if self.stage:
        return Comm.exchange_rows_start(self, rows, send_counts, recv_counts)
    sc, rc = [int(c) for c in send_counts], [int(c) for c in recv_counts]
    rows = rows.contiguous()
    out = torch.empty((sum(rc),) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    work = self.dist.all_to_all_single(out, rows, output_split_sizes=rc, input_split_sizes=sc, group=self.group, async_op=True)

    def wait():
        st = stats()
        if st.profile:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            work.wait()
            e1.record()
            st.wait_events.append((e0, e1))
        else:
            work.wait()  # (``rows`` / ``out`` stay referenced by this closure until then)
        return out
    return wait


TorchDistComm.exchange_rows_start = _torchdist_exchange_rows_start


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

    # (the two exchanges a step repeats stand in for ONE collective each: one gathering copy, not a copy per peer -- the
    # dispatch counts of profiles/*virtual_rank_kernel_time.md are read as those of the RCCL path)
    def exchange_rows_start(self, rows, send_counts, recv_counts):
        hub = self.hub
        starts = [0]
        for c in send_counts:
            starts.append(starts[-1] + int(c))
        hub.slots[self.rank] = (rows, starts)
        hub.barrier.wait()
        parts = []
        for r in range(self.world):
            src, st = hub.slots[r]
            if st[self.rank + 1] > st[self.rank]:
                parts.append(src[st[self.rank]:st[self.rank + 1]])
        out = torch.cat(parts, dim=0) if parts else rows[:0].clone()
        hub.barrier.wait()
        return lambda: out

    def exchange_counts(self, counts):
        world, k = counts.shape
        hub = self.hub
        hub.slots[self.rank] = counts
        hub.barrier.wait()
        recv = torch.stack([hub.slots[r][self.rank] for r in range(world)])
        hub.barrier.wait()
        both = host(torch.cat([counts.reshape(-1), recv.reshape(-1)]))
        n = world * k
        return ([both[r * k:(r + 1) * k] for r in range(world)], [both[n + r * k:n + (r + 1) * k] for r in range(world)])


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
class BlockDecomposition:
    """Axis-aligned blocks (8 ranks = 2x2x2, 4 = 2x2x1, 2 = 2x1x1): ``cuts[k]`` = the increasing
    inner cut planes along axis k, rank = (ix * ny + iy) * nz + iz.  Region of a rank = the product of its per-axis
    intervals [cut[i-1], cut[i]) with -inf / +inf at the ends: ownership is a pure function of the position, per axis, so
    every point -- particle or lattice point -- has exactly one owner."""

    def __init__(self, cuts):
        self.cuts = [[float(c) for c in ax] for ax in cuts]
        assert len(self.cuts) == 3
        for ax in self.cuts:
            assert all(a < b for a, b in zip(ax, ax[1:])), "cuts must increase"
        self.grid = [len(ax) + 1 for ax in self.cuts]
        self.world = self.grid[0] * self.grid[1] * self.grid[2]

    @staticmethod
    def uniform(lo, hi, grid):
        return BlockDecomposition([[lo[k] + (hi[k] - lo[k]) * i / grid[k] for i in range(1, grid[k])] for k in range(3)])

    @staticmethod
    def medians(pos, grid):
        """Cut planes at particle-count quantiles of a (host) position array, axis by axis (load balance)."""
        cuts = []
        for k in range(3):
            q = [i / grid[k] for i in range(1, grid[k])]
            cuts.append([float(v) for v in np.quantile(np.asarray(pos)[:, k], q)] if q else [])
        return BlockDecomposition(cuts)

    def coords(self, r):
        nx, ny, nz = self.grid
        return (r // (ny * nz), (r // nz) % ny, r % nz)

    def bounds(self, r):
        """[(lo, hi)] per axis of rank r's block."""
        out = []
        for k, i in enumerate(self.coords(r)):
            ax = self.cuts[k]
            out.append((-float("inf") if i == 0 else ax[i - 1], float("inf") if i == len(ax) else ax[i]))
        return out

    def owner(self, pos):
        own = torch.zeros(pos.shape[0], dtype=torch.int64, device=pos.device)
        for k in range(3):
            if self.grid[k] > 1:
                cache = self.__dict__.setdefault("_cuts_cache", {})
                b = cache.get((k, pos.dtype, str(pos.device)))
                if b is None:  # (a tensor from Python numbers is a blocking host -> device copy)
                    b = cache[(k, pos.dtype, str(pos.device))] = torch.tensor(self.cuts[k], dtype=pos.dtype, device=pos.device)
                i = torch.bucketize(pos[:, k].contiguous(), b, right=True)
            else:
                i = 0
            own = own * self.grid[k] + i
        return own

    def gap2(self, pos, r):
        """Squared distance of each point to the block of rank r (0 inside)."""
        d2 = torch.zeros(pos.shape[0], dtype=pos.dtype, device=pos.device)
        for k, (lo, hi) in enumerate(self.bounds(r)):
            if lo == -float("inf") and hi == float("inf"):
                continue
            x = pos[:, k]
            g = torch.clamp(torch.maximum(lo - x, x - hi), min=0.0)
            d2 = d2 + g * g
        return d2

    def within(self, pos, r, width):
        """mask of points whose distance to the block of rank r is <= width (points on the upper faces count as outside
        the half-open block, distance 0: they are within any width)."""
        return self.gap2(pos, r) <= float(width) * float(width)

    def neighbours(self, r, width):
        """Ranks whose block is within ``width`` of rank r's block (the only possible ghost peers)."""
        out = []
        mine = self.bounds(r)
        for q in range(self.world):
            if q == r:
                continue
            d2 = 0.0
            for (alo, ahi), (blo, bhi) in zip(mine, self.bounds(q)):
                g = max(blo - ahi, alo - bhi, 0.0)
                d2 += g * g
            if d2 <= width * width:
                out.append(q)
        return out


class SlabDecomposition(BlockDecomposition):
    """Slabs along one axis: the one-axis special case of :class:`BlockDecomposition` (cuts[0] = -inf, cuts[world] = +inf)."""

    def __init__(self, axis, inner_cuts):
        self.axis = int(axis)
        self.inner = [float(c) for c in inner_cuts]
        cuts = [[], [], []]
        cuts[self.axis] = self.inner
        super().__init__(cuts)

    @staticmethod
    def uniform(axis, lo, hi, world):
        return SlabDecomposition(axis, [lo + (hi - lo) * r / world for r in range(1, world)])


# DMCF_SHARD_FORCE_COMM=1: a single rank still runs every feature exchange (of zero ghost rows) -- the only way to drive the
# asynchronous RCCL calls of the exchange on a 1-GPU box (tests/test_gpu_parallel.py)
FORCE_COMM = os.environ.get("DMCF_SHARD_FORCE_COMM") == "1"


def _bounds_tensor(decomp, ranks, device):
    """float32 [len(ranks), 3, 2]: (lo, hi) per axis of the blocks of ``ranks`` (kept per decomposition: a tensor made from
    Python numbers is a blocking host -> device copy, and a step asks for the same few dozens of times)."""
    cache = decomp.__dict__.setdefault("_bounds_cache", {})
    key = (tuple(ranks), str(device))
    t = cache.get(key)
    if t is None:
        t = cache[key] = torch.tensor([[[lo, hi] for lo, hi in decomp.bounds(r)] for r in ranks], dtype=torch.float32, device=device)
    return t


def _boxes_tensor(decomp, ranks, device):
    """float32 [len(ranks), 6]: lo x, y, z, hi x, y, z of the blocks of ``ranks`` (the layout dmcf_ghost_count takes)."""
    t = _bounds_tensor(decomp, ranks, device)
    cache = decomp.__dict__.setdefault("_boxes_cache", {})
    key = (tuple(ranks), str(device))
    b = cache.get(key)
    if b is None:
        b = cache[key] = torch.cat([t[:, :, 0], t[:, :, 1]], dim=1).contiguous()
    return b


def _gap2_all(pos, bounds):
    """Squared distances of every pos[i] to every block of ``bounds`` ([P, 3, 2]) -> [P, n]; per pair the arithmetic of _gap2."""
    g = torch.clamp(torch.maximum(bounds[:, None, :, 0] - pos[None], pos[None] - bounds[:, None, :, 1]), min=0.0)
    g = g * g
    return (g[..., 0] + g[..., 1]) + g[..., 2]


def _gap2(pos, bounds_rows):
    """Squared distance of pos[i] to the block whose bounds are bounds_rows[i] ([m, 3, 2]); same arithmetic per point as
    BlockDecomposition.gap2 (sender and receiver must decide identically on bit-identical copies of a position)."""
    g = torch.clamp(torch.maximum(bounds_rows[:, :, 0] - pos, pos - bounds_rows[:, :, 1]), min=0.0)
    g = g * g
    return (g[:, 0] + g[:, 1]) + g[:, 2]


class GhostPlan:
    """Ghost copies of one owned point set for one halo width: built once per step, reused by every layer.

    A WIDE plan (``parent`` None) selects, for every rank whose block is within ``width`` of this rank's, the owned points
    within ``width`` of that block (squared distance to the box), and exchanges their positions once: two host round trips
    (the selection sizes, the receive counts).  A NARROW plan derives from a wide one of the same point set WITHOUT
    communication: the sender keeps the rows of the wide send lists that pass the narrower test, the receiver runs the same
    test on the bit-identical received positions and gets the same rows in the same order."""

    def __init__(self, comm, decomp, pos_owned, width, parent=None):
        self.comm, self.decomp = comm, decomp
        self.width = float(width) * (1.0 + 1e-5) + 1e-6  # superset slack; the search re-tests distances exactly
        # The cut planes live in SCENE coordinates; a model with a transformation (models/pbf_model.py:252-301: translate,
        # scale, the rotation of grav_eqvar) convolves transformed positions.  ``decomp.view`` maps those back for the
        # ownership and ghost tests, ``decomp.inflate`` widens the halo by the largest shrink factor of the scale so that it
        # still holds everything within ``width`` in the model's metric (ShardedSimulator.begin sets both).
        view = getattr(decomp, "view", None)
        self.test_pos = view(pos_owned) if view is not None else pos_owned
        self.test_width = self.width * float(getattr(decomp, "inflate", 1.0))
        self.n_owned = pos_owned.shape[0]
        dev = pos_owned.device
        world, rank = comm.world, comm.rank
        empty = torch.zeros(0, dtype=torch.int64, device=dev)
        self.send_idx = [empty] * world
        self.recv_counts = [0] * world
        self.in_parent = None
        self.parent = parent
        w2 = self.test_width * self.test_width
        if world == 1:
            self.ghost_pos = pos_owned[:0]
            self.in_parent = empty
        elif parent is None:
            peers = decomp.neighbours(rank, self.test_width)
            if peers and self.n_owned:
                # candidates: owned points within ``width`` of one of the block's own finite faces
                near = torch.zeros(self.n_owned, dtype=torch.bool, device=dev)
                for k, (lo, hi) in enumerate(decomp.bounds(rank)):
                    x = self.test_pos[:, k]
                    if lo != -float("inf"):
                        near |= (x - lo) <= self.test_width
                    if hi != float("inf"):
                        near |= (hi - x) <= self.test_width
                cand = torch.nonzero(near).reshape(-1)
                cpos = self.test_pos[cand]
                flags = _gap2_all(cpos, _bounds_tensor(decomp, peers, dev)) <= w2   # [P, candidates]
                hit = torch.nonzero(flags)                                  # rows ordered by peer, then by point
                counts = host(torch.bincount(hit[:, 0], minlength=len(peers)))
                rows = cand[hit[:, 1]]
                off = 0
                for i, r in enumerate(peers):
                    self.send_idx[r] = rows[off:off + counts[i]]
                    off += counts[i]
            recv = comm.all_to_all([pos_owned[i] for i in self.send_idx])
            self.recv_counts = [int(r.shape[0]) for r in recv]
            self.ghost_pos = torch.cat(recv, dim=0)
        else:
            counts = self._derive_begin(pos_owned)
            self._derive_finish(host(torch.cat(counts)) if counts else [])
        if parent is None or world == 1:
            self._finish_ext(pos_owned)

    @staticmethod
    def build_fused(comm, decomp, pos_owned, widths):
        """ALL ghost plans of one owned point set -- its widest and every narrower one the step will ask for -- with two
        selection kernels per side (dmcf_ghost_count / dmcf_ghost_write, csrc/ghost.hip) and ONE host round trip, instead of
        ~40 torch calls and two round trips for the widest plan plus ~10 calls per derived one.  Sender: the squared gaps of the
        owned points to the peers' blocks against all widths at once; the per-width counts travel in one collective together with
        what the peers announce (Comm.exchange_counts); receiver: the same kernels over the copies it received, against its own
        block, give the narrower plans' positions inside the widest plan's ghosts.  Lists, orders and ghost sets are those of the
        host form (wide plan + derived plans): DMCF_SHARD_CHECK=1 compares.  -> {round(width, 9): plan}."""
        keys = sorted({round(float(w), 9) for w in widths}, reverse=True)
        dev = pos_owned.device
        world, rank = comm.world, comm.rank
        view = getattr(decomp, "view", None)
        inflate = float(getattr(decomp, "inflate", 1.0))
        test_pos = view(pos_owned) if view is not None else pos_owned
        empty = torch.zeros(0, dtype=torch.int64, device=dev)
        plans = []
        for w in keys:
            p = GhostPlan.__new__(GhostPlan)
            p.comm, p.decomp = comm, decomp
            p.width = float(w) * (1.0 + 1e-5) + 1e-6
            p.test_pos = test_pos
            p.test_width = p.width * inflate
            p.n_owned = pos_owned.shape[0]
            p.send_idx = [empty] * world
            p.recv_counts = [0] * world
            p.parent = plans[0] if plans else None
            p.in_parent = None
            plans.append(p)
        wide, W = plans[0], len(plans)
        w2 = [p.test_width * p.test_width for p in plans]
        peers = decomp.neighbours(rank, wide.test_width)
        announce = torch.zeros((world, W), dtype=torch.int64, device=dev)
        sel = None
        if peers and wide.n_owned:
            sel = ops.ghost_select(test_pos, _boxes_tensor(decomp, peers, dev), w2)
            cache = decomp.__dict__.setdefault("_boxes_cache", {})
            at = cache.get(("peers", tuple(peers), str(dev)))
            if at is None:
                at = cache[("peers", tuple(peers), str(dev))] = torch.tensor(peers, dtype=torch.int64, device=dev)
            announce.index_copy_(0, at, sel.totals.t().contiguous())
        sent, announced = comm.exchange_counts(announce)  # (the one host round trip)
        if sel is not None:
            lists = sel.write([sum(int(sent[r][wi]) for r in peers) for wi in range(W)])
            for wi, p in enumerate(plans):
                off = 0
                for r in peers:
                    c = int(sent[r][wi])
                    p.send_idx[r] = lists[wi][off:off + c]
                    off += c
        for wi, p in enumerate(plans):
            p.recv_counts = [int(announced[r][wi]) for r in range(world)]
        recv = comm.all_to_all([pos_owned[i] for i in wide.send_idx], recv_counts=wide.recv_counts)
        g = wide.ghost_pos = torch.cat(recv, dim=0)
        if W > 1:
            if g.shape[0]:
                gsel = ops.ghost_select(view(g) if view is not None else g, _boxes_tensor(decomp, [rank], dev), w2)
                glists = gsel.write([sum(p.recv_counts) for p in plans])
            for wi, p in enumerate(plans[1:], 1):
                p.in_parent = glists[wi] if g.shape[0] else empty
                p.ghost_pos = g[p.in_parent] if g.shape[0] else g
        for p in plans:
            p._finish_ext(pos_owned)
        return dict(zip(keys, plans))

    def _finish_ext(self, pos_owned):
        self.pos_ext = torch.cat([pos_owned, self.ghost_pos], dim=0).contiguous() if self.comm.world > 1 else pos_owned

    # A NARROW plan in two halves, so that the plans of a whole step can share ONE host round trip (derive_batch): everything
    # that runs on the device first, then -- with the per-peer row counts on the host -- the slicing into per-peer lists.
    def _derive_begin(self, pos_owned):
        parent, decomp, comm = self.parent, self.decomp, self.comm
        dev = pos_owned.device
        world, rank = comm.world, comm.rank
        assert parent.parent is None and parent.n_owned == self.n_owned and self.width <= parent.width
        w2 = self.test_width * self.test_width
        self._pos_owned = pos_owned
        view = getattr(decomp, "view", None)
        counts = []
        # What does not depend on the narrower width is formed once per wide plan: its send rows in one piece with their peer
        # number and their squared distance to that peer's block, and the same for the received copies and this rank's block.
        geo = parent.__dict__.get("_derive_geo")
        if geo is None:
            peers = [r for r in range(world) if parent.send_idx[r].shape[0] > 0]
            rows = seg = sgap = None
            if peers:
                rows = torch.cat([parent.send_idx[r] for r in peers])
                seg = torch.repeat_interleave(torch.arange(len(peers), device=dev),
                                              torch.tensor([parent.send_idx[r].shape[0] for r in peers], device=dev))
                sgap = _gap2(parent.test_pos[rows], _bounds_tensor(decomp, peers, dev)[seg])
            g = parent.ghost_pos
            src = rgap = None
            if g.shape[0]:
                src = torch.repeat_interleave(torch.arange(world, device=dev), torch.tensor(parent.recv_counts, device=dev))
                rgap = _gap2(view(g) if view is not None else g, _bounds_tensor(decomp, [rank], dev).expand(g.shape[0], 3, 2))
            geo = parent.__dict__["_derive_geo"] = (peers, rows, seg, sgap, src, rgap)
        self._peers, rows, seg, sgap, src, rgap = geo
        # sender side: the rows of the wide lists that are within the narrower width of the peer's block
        if self._peers:
            keep = sgap <= w2
            counts.append(torch.bincount(seg[keep], minlength=len(self._peers)))
            self._rows = rows[keep]
        # receiver side: the same test on the received copies
        g = parent.ghost_pos
        self._recv = g.shape[0] > 0
        if self._recv:
            keep = rgap <= w2
            self.in_parent = torch.nonzero(keep).reshape(-1)
            counts.append(torch.bincount(src[keep], minlength=world))
            self.ghost_pos = g[self.in_parent]
        else:
            self.in_parent = torch.zeros(0, dtype=torch.int64, device=dev)
            self.ghost_pos = g
        return counts

    def _derive_finish(self, counts):
        """``counts``: the host copy of what _derive_begin returned, concatenated."""
        counts = [int(c) for c in counts]
        off = 0
        if self._peers:
            c, counts = counts[:len(self._peers)], counts[len(self._peers):]
            for i, r in enumerate(self._peers):
                self.send_idx[r] = self._rows[off:off + c[i]]
                off += c[i]
            del self._rows
        if self._recv:
            self.recv_counts = counts[:self.comm.world]
        self._finish_ext(self._pos_owned)
        del self._pos_owned

    @staticmethod
    def derive_batch(requests):
        """[(wide plan, its owned positions, width), ...] -> the narrow plans, with ONE device -> host read for all of them
        (each plan alone costs one: a step of the Liquid3d net derives ~10)."""
        plans, pending, sizes = [], [], []
        for wide, pos_owned, width in requests:
            p = GhostPlan.__new__(GhostPlan)
            p.comm, p.decomp, p.parent = wide.comm, wide.decomp, wide
            p.width = float(width) * (1.0 + 1e-5) + 1e-6
            p.test_pos = wide.test_pos
            p.test_width = p.width * float(getattr(wide.decomp, "inflate", 1.0))
            p.n_owned = pos_owned.shape[0]
            empty = torch.zeros(0, dtype=torch.int64, device=pos_owned.device)
            p.send_idx = [empty] * wide.comm.world
            p.recv_counts = [0] * wide.comm.world
            c = p._derive_begin(pos_owned)
            plans.append(p)
            pending.extend(c)
            sizes.append(sum(int(t.shape[0]) for t in c))
        flat = host(torch.cat(pending)) if pending else []
        off = 0
        for p, n in zip(plans, sizes):
            p._derive_finish(flat[off:off + n])
            off += n
        return plans

    def extend(self, feats_owned):
        """[n_owned, C] -> [n_owned + n_ghost, C] (owned rows first, ghosts in the order of ``pos_ext``): one all-to-all-v,
        no host round trip (the counts are the plan's)."""
        if self.comm.world == 1 and not FORCE_COMM:
            return feats_owned
        return self.extend_start(feats_owned)()

    def extend_start(self, feats_owned):
        """:meth:`extend`, started now and finished by the returned function (the exchange overlaps whatever is enqueued in
        between)."""
        if self.comm.world == 1 and not FORCE_COMM:
            return lambda: feats_owned
        cat = self.__dict__.get("_send_cat")
        if cat is None:
            cat = self._send_cat = (torch.cat(self.send_idx), [int(i.shape[0]) for i in self.send_idx])
        wait = self.comm.exchange_rows_start(feats_owned[cat[0]], cat[1], self.recv_counts)
        return lambda: torch.cat([feats_owned, wait()], dim=0)

    def extend_from(self, wide, wide_ext):
        """``extend`` without communication, from the same features already extended by a wider plan of the same point set:
        the set's widest plan (this one's parent) or another plan derived from it -- the narrower test keeps a subset of the
        wider one's ghosts, in the same order."""
        ghosts = wide_ext[self.n_owned:]
        if wide is self.parent:
            ghosts = ghosts[self.in_parent]
        else:
            assert wide.parent is self.parent and wide.width >= self.width
            ghosts = ghosts[torch.searchsorted(wide.in_parent, self.in_parent)]
        return torch.cat([wide_ext[:self.n_owned], ghosts], dim=0)


# --------------------------------------------------------------------------------------------------
# sharded per-step driver for the PBFNet family (SymNet / HRNet / CConv)
# --------------------------------------------------------------------------------------------------
class ShardedSimulator:
    """``step(state) -> state`` on the particles this rank owns.

    state = dict(pos, vel, acc|None, box, box_normals, gid): fluid particles this rank holds (``gid`` = global particle
    id, int64, carried through migration; a step starts by handing every particle to the owner of its advected position and
    returns the particles it then computed, so between steps a particle may sit slightly outside its holder's block), owned
    boundary particles (static).  The network is the model's OWN
    ``forward`` (models/hrnet.py, sym_net.py, cconv.py): this class only installs ``model.conv_hook``, through which
    every ContinuousConv call of the forward pass gets its input rows extended by the ghosts within the layer's radius."""

    def __init__(self, model, comm, decomp, reserve_gib=None):
        self.model = model
        self.comm = comm
        self.decomp = decomp
        assert decomp.world == comm.world
        self.reserve_gib = reserve_gib  # as Simulator(reserve_gib=...): one block for the caching allocator before the first step
        self.reserved_gib = None
        self.exchanged_rows = 0
        self.migrated_rows_total = 0
        self.host_syncs_last_step = None  # device -> host reads of the last step (each one drains the queue): see Stats
        self.launch_rows = []             # (point set, radius, input rows of the launch, owned rows) per convolution of the last step
        m = model
        import copy
        # the decomposition the ghost plans test against: the caller's cut planes (scene coordinates) seen through the model's
        # transformation (translate / scale / grav_eqvar, pbf_model.py:252-301) -- a copy per simulator, its view is set per step
        self._mdecomp = copy.copy(decomp)
        self._mdecomp.__dict__.pop("_bounds_cache", None)
        self._mdecomp.__dict__.pop("_boxes_cache", None)
        if m.dens_feats or m.pres_feats or m.dens_norm or m.use_pre_adv or m.use_feats:
            raise NotImplementedError("dens_feats / pres_feats / dens_norm / use_pre_adv / use_feats in the sharded path")
        if not m.use_bnds and type(m).__name__ == "SymNet":
            raise NotImplementedError("use_bnds=False with the ASCC head in the sharded path")
        if m.voxel_size is None and any(s != 1 for s in m.strides):
            raise NotImplementedError("FPS based multi-scale (voxel_size None) in the sharded path: farthest point sampling "
                                      "is sequential over the whole point set")

    # -- helpers -----------------------------------------------------------------------------------
    def _plan(self, name, width):
        """The ghost plan of point set ``name`` for ``width``: derived without communication from the set's widest plan
        (built when the set is registered, at the largest radius of the network)."""
        key = (name, round(float(width), 9))
        plan = self._plans.get(key)
        if plan is not None and key in getattr(self, "_unregistered", ()):
            self._unregistered.discard(key)
            self._register_lattice(name, plan)
        if plan is None:
            wide = self._wide[name]
            if float(width) * (1.0 + 1e-5) + 1e-6 > wide.width:
                raise RuntimeError(f"ghost plan of {name!r} asked for width {width} > the step's widest {wide.width}")
            plan = GhostPlan(self.comm, self._mdecomp, self._sets[name], width, parent=wide)
            if os.environ.get("DMCF_SHARD_CHECK") == "1" and self.comm.world > 1:
                direct = GhostPlan(self.comm, self._mdecomp, self._sets[name], width)
                if not torch.equal(direct.ghost_pos, plan.ghost_pos) or any(
                        not torch.equal(a, b) for a, b in zip(direct.send_idx, plan.send_idx)):
                    raise RuntimeError("a derived ghost plan differs from the directly built one")
            self._plans[key] = plan
            self._register_lattice(name, plan)
        return plan

    def _add_set(self, name, pos, width):
        """Register an owned point set and build its widest ghost plan (the only one that communicates)."""
        self._sets[name] = pos
        self._name_of[id(pos)] = name
        fused = os.environ.get("DMCF_SHARD_FUSED", "1")  # "0": the host form below; "force": also for CPU tensors (tests/shims.py)
        # (dmcf_ghost_count takes at most 64 peer boxes per call: beyond 65 ranks -- the same decision on every rank, a collective
        # follows -- the plans come from the host form below, one width at a time)
        if (self.comm.world > 1 or FORCE_COMM) and self.comm.world <= 65 and (fused == "force" or (fused != "0" and pos.is_cuda)):
            # every width the step will ask this set for is configuration: the layers' radii and, for the particles, the halos
            # the lattices are built from -- all plans at once (GhostPlan.build_fused)
            m = self.model
            widths = [float(width)] + [float(np.float32(r)) for r in m.particle_radii
                                       if float(r) * (1.0 + 1e-5) + 1e-6 <= float(width) * (1.0 + 1e-5) + 1e-6]
            if name == "s0":
                widths += [self._lattice_margin(st) for st in m.strides if st != 1]
            if len({round(float(w), 9) for w in widths}) > 8:  # (dmcf_ghost_count takes 8 widths: the rarer ones are derived)
                widths = sorted(widths, reverse=True)[:1] + sorted({float(np.float32(r)) for r in m.particle_radii}, reverse=True)[:7]
            plans = GhostPlan.build_fused(self.comm, self._mdecomp, pos, widths)
            wide = plans[round(float(width), 9)]
            for w, plan in plans.items():
                self._plans[(name, w)] = plan
                if os.environ.get("DMCF_SHARD_CHECK") == "1":
                    direct = GhostPlan(self.comm, self._mdecomp, pos, w)
                    if not torch.equal(direct.ghost_pos, plan.ghost_pos) or any(
                            not torch.equal(a, b) for a, b in zip(direct.send_idx, plan.send_idx)):
                        raise RuntimeError("a fused ghost plan differs from the directly built one")
            self._wide[name] = wide
            self._unregistered = getattr(self, "_unregistered", set()) | {(name, w) for w, q in plans.items() if q is not wide}
            return wide
        wide = GhostPlan(self.comm, self._mdecomp, pos, width)
        self._wide[name] = wide
        self._plans[(name, round(float(width), 9))] = wide
        return wide

    def _register_lattice(self, name, plan):
        lat = self._lattices.get(name)
        if lat is not None:
            lattice.register_points(plan.pos_ext, lat[0], lat[1], ("sharded", lat[0].data_ptr()), lat[2], center_host=lat[3])

    def _lattice_margin(self, stride):
        """Halo of particles a rank needs to build every lattice point of ``grid_pos`` (losses.py:136-181) it owns: a point
        is a corner (+- pad) of a voxel holding a particle, with +- hyst hysteresis, i.e. within (1 + hyst + pad) voxels of
        the particle PER AXIS -- the ghost test measures the Euclidean distance to the block, hence the sqrt(3)."""
        m = self.model
        vs = np.asarray(m.voxel_size, dtype=np.float32) * np.float32(stride)
        return float(vs.max()) * (1.0 + m.sample_hyst + m.sample_pad + 0.05) * 3.0 ** 0.5

    def _conv_hook(self, conv, feats, inp_pos, out_pos, extent, widest_extent=None):
        """model.conv_hook: conv(feats, inp_pos -> out_pos) with the input rows extended by the ghosts within extent / 2.
        ``widest_extent``: the largest extent any layer reading the SAME ``feats`` uses -- the ghost rows then travel once,
        at that width, and the narrower sets are subsets of those rows (GhostPlan.extend_from)."""
        inp = self._set_name(inp_pos)
        plan = self._plan(inp, 0.5 * float(extent))
        n_own = feats.shape[0]
        if self.comm.world == 1 and not FORCE_COMM:
            ext = feats
        else:
            wider = widest_extent is not None and float(widest_extent) > float(extent)
            wide = self._plan(inp, 0.5 * float(widest_extent)) if wider else plan
            hit = self._shared.get(id(feats))
            if hit is not None and (hit[0] is not feats or hit[1] is not wide):
                hit = None
            if hit is not None and callable(hit[2]):
                hit = (feats, wide, hit[2]())  # started by _ghost_prefetch when the layer began: finish it now
                self.exchanged_rows += hit[2].shape[0] - n_own
                self._shared[id(feats)] = hit
            if hit is None:
                hit = (feats, wide, wide.extend(feats))
                self.exchanged_rows += hit[2].shape[0] - n_own
                if wider:  # other layers will read the same rows at their own widths
                    while len(self._shared) >= 4:
                        self._shared.pop(next(iter(self._shared)))
                    self._shared[id(feats)] = hit
            ext = hit[2] if plan is wide else plan.extend_from(wide, hit[2])
        # (the launch reads the ghosts within ITS radius, whatever width the rows travelled at)
        self.launch_rows.append((inp, 0.5 * float(extent), int(plan.pos_ext.shape[0]), int(n_own)))
        return conv(ext, plan.pos_ext, out_pos, extent, None)

    def _ghost_prefetch(self, requests):
        """model.ghost_prefetch: a layer is about to read these (features, input positions, widest extent) -- start the ghost
        exchange of ALL of them now.  The first convolution of the layer waits for its own exchange only; the others travel
        (RCCL's stream) while it computes."""
        if (self.comm.world == 1 and not FORCE_COMM) or os.environ.get("DMCF_SHARD_PREFETCH", "1") == "0":
            return
        for feats, inp_pos, widest_extent in requests:
            wide = self._plan(self._set_name(inp_pos), 0.5 * float(widest_extent))
            if id(feats) not in self._shared:
                while len(self._shared) >= 4:
                    self._shared.pop(next(iter(self._shared)))
                self._shared[id(feats)] = (feats, wide, wide.extend_start(feats))

    # -- one step ----------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, state):
        """One time step.  Like Simulator.run_inference the searches run with row capacities estimated from the previous
        step (single pass, no host round trips); whether any rank outgrew an estimate is agreed with ONE tiny all-reduce
        after the step -- the validation happens when the cache scope closes, i.e. after every collective of the step --
        and then all ranks repeat the step with the exact search."""
        if self.reserved_gib is None:
            from .pipelines.simulator import reserve_for_scene
            self.reserved_gib = 0.0
            if state["pos"].is_cuda:
                self.reserved_gib = reserve_for_scene(self.reserve_gib, int(state["pos"].shape[0] + state["box"].shape[0]),
                                                      state["pos"].device)
        syncs0 = stats().host_syncs
        self.launch_rows = []
        try:
            with neighbor_cache(estimate=True):
                out = self._step(state)
            ok = 1
        except ops.NeighborCapacityExceeded:
            out, ok = None, 0
        flag = torch.tensor([ok], dtype=torch.int32, device=state["pos"].device)
        self.comm.all_reduce(flag, "min")
        if int(host(flag[0])) == 0:
            with neighbor_cache(estimate=False):
                out = self._step(state)
        self.host_syncs_last_step = stats().host_syncs - syncs0
        return out

    def _step(self, state):
        """Migration, then the model's OWN five stages (models/base_model.py: transform -> preprocess -> forward -> postprocess
        -> inv_transform) with this object installed as ``model.shard`` / ``model.conv_hook`` / ``model.ghost_prefetch``: the
        model asks it for the three things that need other ranks -- the fluid bounding box of the boundary crop, the lattices of
        the coarser scales, the ghost rows of every convolution's input.  Nothing of PBFNet.preprocess / postprocess is
        re-stated here."""
        m, comm = self.model, self.comm
        self._plans, self._sets, self._lattices, self._wide, self._name_of, self._shared, self._pending = {}, {}, {}, {}, {}, {}, {}
        self._unregistered = set()
        pos0, vel0, acc = state["pos"], state["vel"], state.get("acc")
        gid = state["gid"]
        box_all, bfeats_all = state["box"], state["box_normals"]
        # Migration FIRST, by the ADVECTED positions: the network convolves the integrated positions (pbf_model.py:318), so a
        # rank must own the particles whose integrated position lies in its block -- then every output point of a layer is
        # inside its owner's block and a halo of the layer's radius holds all its neighbours.  (Owning by the positions the
        # step starts from leaves outputs up to dt |v| outside their block, where the neighbours' ghost test -- distance to
        # the BLOCK -- no longer covers them.)  One stable sort by owner, one host round trip (the send counts), two
        # all-to-all-v (payload; global ids with the receive counts the first one produced).
        self.migrated_rows = 0
        if comm.world > 1:
            adv, _ = m.integrate_pos_vel(pos0, vel0, acc)
            payload = torch.cat([pos0, vel0] + ([acc] if acc is not None else []), dim=1)
            fused = os.environ.get("DMCF_SHARD_FUSED", "1")
            if comm.world <= 64 and (fused == "force" or (fused != "0" and adv.is_cuda)):  # (dmcf_ghost_count takes 64 boxes)
                # the selection kernels with the ownership test (csrc/ghost.hip): rows per owner and the stable order by owner
                sel = ops.ghost_select(adv, _boxes_tensor(self.decomp, list(range(comm.world)), adv.device), [-1.0])
                counts = host(sel.totals[0])
                if int(sum(counts)) != adv.shape[0]:
                    raise RuntimeError(f"{adv.shape[0] - int(sum(counts))} particles with a non-finite position: no rank owns them")
                order = sel.write([adv.shape[0]])[0]
                if os.environ.get("DMCF_SHARD_CHECK") == "1":
                    assert torch.equal(order, torch.argsort(self.decomp.owner(adv), stable=True))
            else:
                own = self.decomp.owner(adv)
                order = torch.argsort(own, stable=True)
                counts = host(torch.bincount(own, minlength=comm.world))
            self.migrated_rows = int(sum(counts)) - int(counts[comm.rank])
            self.migrated_rows_total += self.migrated_rows
            recv = comm.all_to_all(list(torch.split(payload[order], counts, dim=0)))
            rc = [int(r.shape[0]) for r in recv]
            payload = torch.cat(recv, dim=0)
            gid = torch.cat(comm.all_to_all(list(torch.split(gid[order], counts, dim=0)), recv_counts=rc), dim=0)
            pos0, vel0 = payload[:, 0:3].contiguous(), payload[:, 3:6].contiguous()
            acc = payload[:, 6:9].contiguous() if acc is not None else None
        m.shard, m.conv_hook, m.ghost_prefetch = self, self._conv_hook, self._ghost_prefetch
        try:
            new_pos, new_vel = m([pos0, vel0, acc, None, box_all, bfeats_all], training=False)
        finally:
            m.shard = m.conv_hook = m.ghost_prefetch = None
        self.net_output, self.pos_correction = m.net_output, m.pos_correction
        return dict(pos=new_pos.contiguous(), vel=new_vel.contiguous(), acc=acc, box=box_all, box_normals=bfeats_all, gid=gid)

    # -- what PBFNet.preprocess asks of a sharded step (model.shard) ----------------------------------------------------------
    def fluid_bounds(self, mn, mx):
        """pbf_model.py:330-334: the fluid bounding box over ALL ranks (one collective for both ends: max(-lo), max(hi))."""
        lohi = self.comm.all_reduce(torch.cat([-mn, mx]), "max")
        return -lohi[:3], lohi[3:]

    def begin(self, all_pos, pos, box):
        """The point sets of the step exist: register the one every layer reads (its widest ghost plan is the only one that
        communicates); fluid-only / boundary-only sets are registered when a convolution first reads them."""
        m = self.model
        self._set_view(all_pos.device)
        filter_extent = [float(np.float32(r) * np.float32(2)) for r in m.particle_radii]
        r_max = 0.5 * filter_extent[-1]
        multi = any(s != 1 for s in m.strides)
        margin = max(self._lattice_margin(s) for s in m.strides) if multi else 0.0
        self._wide_w = max(r_max, margin)  # the widest ghost set any layer (or the lattice construction) of the step needs
        self._pending = {id(pos): ("pos", pos), id(box): ("box", box)}
        self._add_set("s0", all_pos, self._wide_w)

    def _set_view(self, dev):
        """How the positions the model convolves (after PBFNet.transform) map back to the scene coordinates of the cut planes:
        model = ((scene + translate) * scale) @ R, so scene = (model @ R^T) / scale - translate (the inverse the model itself
        applies, pbf_model.py:280-301, with its clamp of the scale).  Distances shrink by at most min(scale): the halo tested in
        scene coordinates is widened by 1 / min(scale) so that it holds every point within the layer's radius in the model's
        metric (extra ghosts are harmless)."""
        tr = self.model.transformation
        d = self._mdecomp
        if not any(k in tr for k in ("translate", "scale", "grav_eqvar")):
            d.view, d.inflate = None, 1.0
            return
        t = torch.tensor(tr.get("translate", [0.0, 0.0, 0.0]), dtype=torch.float32, device=dev)
        sc = [float(v) for v in tr.get("scale", [1.0, 1.0, 1.0])]
        for k in range(3):
            if sc[k] <= 1e-5 and d.grid[k] > 1:
                raise NotImplementedError(f"scale {sc} collapses axis {k}, which the decomposition cuts")
        s = torch.tensor(sc, dtype=torch.float32, device=dev).clamp(min=1e-5)
        R = self.model.R.t().contiguous() if "grav_eqvar" in tr else None

        def view(x):
            # (an explicit row-wise form, not `x @ R`: sender and receiver of a derived ghost plan evaluate this on row sets of
            # different sizes and rely on bit-identical rows -- a BLAS product of a different shape may sum in another order,
            # and one point within an ulp of a plan's width then gives mismatched all-to-all split sizes: ADVICE r04)
            if R is not None:
                x = (x[:, 0:1] * R[0] + x[:, 1:2] * R[1]) + x[:, 2:3] * R[2]
            return x / s - t
        d.view = view
        d.inflate = 1.0 / min([v for v in sc if v > 1e-5] + [1.0]) if any(v < 1.0 for v in sc) else 1.0
        self._transformed = True

    def _set_name(self, pos):
        name = self._name_of.get(id(pos))
        if name is None:
            name, t = self._pending.pop(id(pos))
            self._add_set(name, t, self._wide_w)
        return name

    def dilated_pos(self, base):
        """get_dilated_pos (losses.py:249-284) for a sharded step: the same lattice on every rank (global origin, one 4-double
        all-reduce), each rank keeps the points it owns.  Returns (point sets per stride, FPS index lists = None)."""
        m, comm = self.model, self.comm
        dev = base.device
        base_name = self._set_name(base)
        wide_w = self._wide_w
        multi = any(s != 1 for s in m.strides)
        sets = []
        center = None
        if m.centralize and multi:
            acc64 = torch.cat([base.t().contiguous().double().sum(dim=1),
                               torch.tensor([float(base.shape[0])], dtype=torch.float64, device=dev)])
            acc64 = comm.all_reduce(acc64, "sum")
            center = (acc64[:3] / acc64[3]).to(torch.float32)
            center_host = host(center)
        for si, stride in enumerate(m.strides):
            if stride == 1:
                sets.append(base)
                continue
            vs = np.asarray(m.voxel_size, dtype=np.float32) * np.float32(stride)
            cand = self._plan(base_name, self._lattice_margin(stride)).pos_ext
            g, gbox = grid_pos(cand, vs, centralize=m.centralize, pad=m.sample_pad, hyst=m.sample_hyst, center=center,
                               return_box=True)
            g = g[self._mdecomp.owner(self._mdecomp.view(g) if getattr(self._mdecomp, 'view', None) is not None else g) == comm.rank].contiguous()
            name = f"s{si}"
            if center is not None and g.is_cuda:
                # all lattices of the step share the agreed centre: the layers between them (and their owned + ghost
                # inputs, see _register_lattice) can take the lattice form of ContinuousConv (dmcf_amd/lattice.py).  The
                # ghost copies a layer adds to this set come from other ranks' lattices: the union of all ranks' boxes
                # (one tiny all-reduce) holds owned and ghost points alike.  Only the INPUT volumes are that large
                # (zero-filled, 4 B x Cin per cell); the kernel walks the output box.  A rank without candidates
                # contributes nothing to the union.
                # gbox is None for a rank whose candidate set is empty (it then has no points of this lattice and joins any
                # union) -- or whose grid_pos took the sort-based form (bounding box too sparse): its points are NOT known
                # to lie in the others' boxes, so the last entry vetoes the lattice form on every rank
                big_i = 1 << 40
                unknown = 1 if (gbox is None and g.shape[0] > 0) else 0
                if gbox is not None:
                    blo, bdims = list(gbox[0]), list(gbox[1])
                    ext = [v for k in range(3) for v in (-blo[k], blo[k] + bdims[k] - 1)]
                else:
                    ext = [-big_i] * 6
                ext = host(comm.all_reduce(torch.tensor(ext + [unknown], dtype=torch.int64, device=dev), "max"))
                if ext[0] > -big_i and ext[6] == 0:
                    ulo = [-ext[2 * k] for k in range(3)]
                    uhi = [ext[2 * k + 1] for k in range(3)]
                    # THIS rank's part of the union: the cells within the widest ghost width of its block (owned points lie
                    # in the block, ghosts within that distance of it); the union only bounds the outer, open sides.  The
                    # dense volumes of the lattice form are zero-filled and walked box by box: with the union box of eight
                    # ranks the four lattice layers took 3x the time of the single-rank step (1.8 against 0.6 ms each).
                    # (model coordinates = scene coordinates only without a transformation: otherwise the union box stays)
                    for k, (b_lo, b_hi) in enumerate(self.decomp.bounds(comm.rank) if getattr(self._mdecomp, 'view', None) is None else []):
                        v = float(vs[k])
                        if v > 1e-5 and b_lo != -float("inf"):
                            ulo[k] = max(ulo[k], int(np.floor((b_lo - wide_w - center_host[k]) / v)) - 2)
                        if v > 1e-5 and b_hi != float("inf"):
                            uhi[k] = min(uhi[k], int(np.ceil((b_hi + wide_w - center_host[k]) / v)) + 2)
                    udims = [max(uhi[k] - ulo[k] + 1, 1) for k in range(3)]
                    self._lattices[name] = (center, [float(v) for v in vs], (ulo, udims), center_host)
                    lattice.register_points(g, center, vs, ("sharded", center.data_ptr()), (ulo, udims),
                                            center_host=center_host)
            self._add_set(name, g, wide_w)
            self._register_lattice(name, self._wide[name])
            sets.append(g)
        # every ghost plan the layers will ask for, NOW: a derived plan costs two small host round trips (its selection
        # sizes), and here the queue is short -- inside the forward pass each would drain it
        if comm.world > 1:
            want = []
            for name in list(self._sets):
                for r in m.particle_radii:
                    w = float(np.float32(0.5) * (np.float32(r) * np.float32(2)))
                    if float(r) * (1.0 + 1e-5) + 1e-6 <= self._wide[name].width and (name, round(w, 9)) not in self._plans \
                            and (name, round(w, 9)) not in [(n, round(x, 9)) for n, x in want]:
                        want.append((name, w))
            for (name, w), plan in zip(want, GhostPlan.derive_batch([(self._wide[name], self._sets[name], w) for name, w in want])):
                if os.environ.get("DMCF_SHARD_CHECK") == "1":
                    direct = GhostPlan(self.comm, self._mdecomp, self._sets[name], w)
                    if not torch.equal(direct.ghost_pos, plan.ghost_pos) or any(
                            not torch.equal(a, b) for a, b in zip(direct.send_idx, plan.send_idx)):
                        raise RuntimeError("a derived ghost plan differs from the directly built one")
                self._plans[(name, round(w, 9))] = plan
                self._register_lattice(name, plan)
        return sets, None


def shard_scene(scene, decomp, rank, device, presharded=False):
    """Owned part of a scene dict(pos, vel, box, box_normals[, acc]) for ``rank`` (numpy in, tensors out).
    ``presharded``: the arrays already are this rank's part (each rank generated only its own block); ownership is still
    checked against the decomposition -- a particle the generator put on the wrong side of a cut would otherwise be
    silently duplicated or lost."""
    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
    pos, box = t(scene["pos"]), t(scene["box"])
    own_p = decomp.owner(pos) == rank
    own_b = decomp.owner(box) == rank
    if presharded and not (bool(own_p.all()) and bool(own_b.all())):
        raise ValueError(f"rank {rank}: pre-sharded scene holds points outside its block")
    state = dict(pos=pos[own_p].contiguous(), vel=t(scene["vel"])[own_p].contiguous(),
                 acc=t(scene["acc"])[own_p].contiguous() if scene.get("acc") is not None else None,
                 box=box[own_b].contiguous(), box_normals=t(scene["box_normals"])[own_b].contiguous(),
                 gid=torch.nonzero(own_p).reshape(-1))
    return state
