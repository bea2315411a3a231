"""ContinuousConv layer -- mirror of the reference's ``utils/convolutions.py:34-473`` on PyTorch-ROCm.

Same constructor keywords (convolutions.py:150-175), same ``call`` signature (:277-286), same side
attributes read by callers (``.nns``, ``._avg_neighbors``, ``._conv_values``, ``._conv_output``), same
weight names (``kernel``, ``bias``), so the model code and configs of the reference drop onto it.
What differs is underneath: the radius search, the window function, the CConv itself and the whole
ASCC body (mirror + two continuous_conv calls + batched matmul, :410-412,:433-458) are one HIP
search + one HIP kernel launch each (dmcf_amd/ops.py -> libdmcf_hip.so).

Inference only: the layer does not record autograd history (training is out of scope, SURVEY.md
section 2 row 16).
"""
import math
import os
import threading

import numpy as np
import torch

from .. import ops
from .tools.losses import WindowFunction

__all__ = ["ContinuousConv", "neighbor_cache"]

# dmcf_cconv_scatter_forward (splat S) is taken for particles -> coarse-lattice layers with these output channel counts (8 is
# implemented and tested but measures slower than the gather kernels: tools/bench_scatter.py) ...
SCATTER_OUT_CHANNELS = (4,)
SCATTER_MIN_INPUTS = 4096           # ... for point sets big enough to fill the device (small scenes are paced by launches),
SCATTER_MAX_LATTICE_CELLS = 1 << 16  # and lattices whose cell coordinates stay exact in float32 arithmetic
SMALL_LIST_ENTRIES = 1 << 21  # padded lists up to this size stay padded whatever their fill (see _NeighborCache.search: 4 total + 2^22)


class _NeighborCache:
    """Per-step reuse of neighbour lists and grids.

    The reference runs its own FixedRadiusSearch inside every ContinuousConv.call
    (convolutions.py:354-358), 18/27/43 searches per step of which only 12/12/19 are distinct
    (SURVEY.md section 3.2).  Inside ``with neighbor_cache():`` a search is keyed by
    (points, queries, radius, ignore_query_point) -- tensors by storage identity -- and computed once;
    the cell-sorted grid of a point set is shared between searches with the same radius.
    Results are identical to searching again (same inputs, deterministic kernel)."""

    def __init__(self):
        self.depth = 0
        # longest-row estimates from the previous step, by WHAT the search is (radius, size class of both point sets, its
        # flags -- _hint_key): with them a step enqueues all its searches without a single host round trip (see search()).
        # (Keyed by the position in the step's sequence of searches through round 3: when a layer between two lattices fell
        # back to a neighbour list in the middle of a rollout, every later search took its neighbour's estimate -- a 35-entry
        # stride for a 286-entry list -- and the step was repeated.)
        self.hints = {}
        self.caps = {}  # (class, n-th search of that class in the step) -> entries of its padded buffers in the previous step
        self.totals = {}  # same keys -> pairs of that list in the previous step (padded rows or CSR: see search())
        self.last_overflow = []  # diagnostics: the lists that outgrew their estimates in the last step that had to be repeated
        # consumers per list: learnt in one step, used in the next to hand a list's buffers back to the allocator as soon
        # as its last consumer has enqueued its kernel (12 lists of 2.4 - 3.4 GB each at 1M particles; all of them alive
        # until the end of the step was 160 GB at 4M particles and sent the caching allocator into free / malloc cycles)
        self.expect = {}
        self.uses = {}
        self.slot_of = {}
        self.done = []
        self.use_hints = False
        self.nth = {}  # searches of each class so far in this step (the second half of a list's key in expect / uses / caps)
        self.pending = []
        self.reports = []  # (int64 device tensor, callback(list) -> outgrown?) read with the step's one synchronisation (report())
        self.lists = {}
        self.tables = {}
        self.plans = {}
        self.keepalive = []
        self.key = None
        self.states = {}  # key -> (hints, caps, expect) of the rollouts that are not the current one

    def select(self, key):
        """Swap in the estimates (hints / caps / expect) of ``key`` (outermost scope only)."""
        if key == self.key:
            return
        self.states[self.key] = (self.hints, self.caps, self.expect, self.totals)
        while len(self.states) > 16:  # a handful of (model, scene) rollouts at a time
            self.states.pop(next(iter(self.states)))
        self.hints, self.caps, self.expect, self.totals = self.states.pop(key, ({}, {}, {}, {}))
        self.key = key

    def __enter__(self):
        if self.depth == 0:
            self.nth = {}
            self.pending = []
            self.reports = []
        self.depth += 1
        return self

    def report(self, values, callback):
        """Device numbers somebody wants on the host when the step ends (int64 tensor; read in the step's single synchronisation,
        no round trip of their own): ``callback(list of ints)`` returns True if they show an estimated capacity outgrown -- the
        step is then repeated like one with an outgrown neighbour list.  Outside a step: read at once."""
        if self.depth == 0:
            return bool(callback(values.flatten().tolist()))
        self.reports.append((values.flatten().long(), callback))
        return False

    def __exit__(self, exc_type, exc, tb):
        self.depth -= 1
        if self.depth == 0:
            pending, self.pending = self.pending, []
            self._flush()
            for _, _, r in pending:  # layers keep their last list in .nns: do not let that pin the buffers into the next step
                r.release()
            if exc_type is None:
                self.expect = dict(self.uses)
            self._maxc = None
            self.uses = {}
            self.slot_of = {}
            from .. import lattice
            lattice.clear()
            self.lists.clear()
            self.tables.clear()
            self.plans.clear()
            self.keepalive.clear()
            reports, self.reports = self.reports, []
            outgrown = False
            if exc_type is None and reports:
                flat = torch.cat([v for v, _ in reports]).tolist()
                at = 0
                for v, cb in reports:
                    outgrown = bool(cb(flat[at: at + v.shape[0]])) or outgrown
                    at += v.shape[0]
            if exc_type is None and outgrown and not pending:
                raise ops.NeighborCapacityExceeded("an estimated capacity was outgrown; repeat the step")
            if exc_type is None and pending:
                # one synchronisation per step: validate the estimated capacities, refresh the estimates (the longest row and
                # the number of pairs of every list)
                # (the padded lists' numbers are gathered by four small launches for all of them, not three per list)
                pad = [k for k, (_, _, r) in enumerate(pending) if isinstance(r, ops.PaddedNeighborList)]
                oth = [k for k, (_, _, r) in enumerate(pending) if not isinstance(r, ops.PaddedNeighborList)]
                parts = []
                if pad:
                    parts.append(torch.cat([pending[k][2].max_count for k in pad]).long())
                    # (a list whose padded rows are small never takes the estimated-CSR form, so its pair count is not needed: no
                    # reduction launch for it -- ten per step of the 2-D models)
                    small = [pending[k][2].row_count.shape[0] * pending[k][2].stride <= SMALL_LIST_ENTRIES for k in pad]
                    zero = ops.const_tensor([0], torch.int64, pending[pad[0]][2].row_count.device)[0]
                    parts.append(torch.stack([zero if sm else pending[k][2].total_ref.long() for k, sm in zip(pad, small)]))
                for k in oth:
                    rs = pending[k][2].neighbors_row_splits
                    parts.append(torch.stack([torch.diff(rs).max() if rs.shape[0] > 1 else rs.new_zeros(()), rs[-1]]))
                flat = torch.cat(parts).tolist()
                vals = [None] * len(pending)
                for i, k in enumerate(pad):
                    vals[k] = (flat[i], None if small[i] else flat[len(pad) + i])
                for i, k in enumerate(oth):
                    vals[k] = (flat[2 * len(pad) + 2 * i], flat[2 * len(pad) + 2 * i + 1])
                fresh, tot = {}, {}
                over = []
                for (hkey, slot, r), (mx, total) in zip(pending, vals):
                    if isinstance(r, ops.PaddedNeighborList):
                        if r.overflowed(mx):
                            over.append((hkey, "row", r.stride, int(mx)))
                    elif r.overflowed(total):
                        over.append((hkey, "pairs", r.capacity, int(total)))
                    fresh[hkey] = max(fresh.get(hkey, 0), int(mx))  # (searches of one class share the longest of their rows)
                    if total is not None:
                        tot[slot] = int(total)
                # estimates come from the PREVIOUS step only: a class this step did not search is forgotten (its next search runs
                # the exact two passes once).  Keeping old entries let one search inherit another's: in the dam break the
                # lattices grow through the half-octave size classes, and at step 81 the s0 -> s2 list (2,300-entry rows)
                # arrived in the class the s2 -> s1 list (300) had left 30 steps earlier -- a repeated step.
                self.hints.clear()
                self.hints.update(fresh)
                self.totals.clear()
                self.totals.update(tot)
                if over or self.use_hints:  # (the repeat of a step runs without estimates: it keeps the record of what was outgrown)
                    self.last_overflow = over
                if over or outgrown:
                    raise ops.NeighborCapacityExceeded("a neighbour list outgrew its estimated capacity; repeat the step")
        return False

    @staticmethod
    def _key(t):
        return (t.data_ptr(), tuple(t.shape), t._version)

    def _max_count_slot(self, device):
        """One element of a tensor zeroed once per step: where a padded search leaves its longest row (a fill per search otherwise)."""
        pool = getattr(self, "_maxc", None)
        if pool is None or pool[1] >= pool[0].shape[0] or pool[0].device != device:
            pool = self._maxc = [torch.zeros(64, dtype=torch.int32, device=device), 0]
        pool[1] += 1
        return pool[0][pool[1] - 1: pool[1]]

    def _flush(self):
        for r in self.done:
            r.release()
        self.done = []

    def _consumer(self, key, res):
        """Count one consumer of the list; after the last one expected the list leaves the cache and its buffers are dropped
        at the next request (by then the consumer -- layers run one after the other -- has enqueued its kernel)."""
        slot = self.slot_of[key]
        self.uses[slot] = self.uses.get(slot, 0) + 1
        if self.expect.get(slot) == self.uses[slot]:
            self.lists.pop(key, None)
            self.done.append(res)

    def scatter_plan(self, inp_positions, out_positions, voxel, radius, block_cells):
        """dmcf_cconv_scatter_plan for this pair of point sets, once per step."""
        key = (self._key(inp_positions), self._key(out_positions), float(radius), int(block_cells))
        plan = self.plans.get(key) if self.depth > 0 else None
        if plan is None:
            plan = ops.scatter_plan(inp_positions, out_positions, voxel, radius, block_cells)
            if self.depth > 0:
                self.plans[key] = plan
        return plan

    def with_query_points(self, frs, points, queries, radius):
        """For a search that ignores the query points (``frs.ignore_query_point``) over points == queries: the list of the
        SAME search with them, if this step already holds one without distances -- the caller's kernel then skips the pairs
        (i, i) itself (DMCF_FLAG_SKIP_SELF) and the second search of the same point set is not run (the ASCC head after the
        trunk's same-scale layers).  Counts as a consumer of that list.  None when there is nothing to share."""
        if self.depth == 0 or not frs.ignore_query_point:
            return None
        points = points.contiguous()
        queries = queries.contiguous()
        if self._key(points) != self._key(queries):
            return None
        key = ((self._key(points), float(radius)), self._key(queries), False, False)
        hit = self.lists.get(key)
        if hit is not None:
            self._flush()
            self._consumer(key, hit)
        return hit

    def search(self, frs, points, queries, radius, distances=True):
        """``distances=False``: the caller evaluates its window in the kernel, which re-forms d^2 from the positions
        (dmcf_hip.h, neighbors_value == NULL) -- inside a step the list then carries no distance array: half the bytes the
        search writes, half the memory of the list."""
        if self.depth == 0:
            return frs(points, queries, radius)
        if not distances and frs.return_distances:
            frs = frs.index_only()
        self._flush()
        points = points.contiguous()
        queries = queries.contiguous()
        tkey = (self._key(points), float(radius))
        key = (tkey, self._key(queries), frs.ignore_query_point, frs.return_distances)
        hit = self.lists.get(key)
        if hit is not None:
            self._consumer(key, hit)
            return hit
        table = self.tables.get(tkey)
        if table is None or table.n_queries_capacity < queries.shape[0]:
            table = ops.build_spatial_hash_table(points, radius, n_queries=max(points.shape[0], queries.shape[0]))
            self.tables[tkey] = table
        hkey = _hint_key(frs, points, queries, radius)
        # what the estimates of a list are filed under -- who consumes it, how big its buffer was -- is what the search IS
        # (its class) plus, for searches of one class within a step, their order among themselves: a layer that switches
        # between the lattice form and a neighbour list mid-rollout then shifts nobody else's entry (as position in the
        # step's whole search sequence did)
        nth = self.nth.get(hkey, 0)
        self.nth[hkey] = nth + 1
        slot = (hkey, nth)
        hint = self.hints.get(hkey) if self.use_hints else None
        # (A class the previous step did not search -- a new layer form, or the same search whose point sets drifted across a size
        # class boundary -- runs the exact two-pass search once.  Borrowing the stride and pair count of a search one class
        # away was tried in two forms (the largest of the candidates; a unique candidate only): it saves that step's host round
        # trip -- 0.2 ms per step of the 1M bench window, where the lattices grow through the classes -- but the borrowed sizes
        # are fresh block sizes for the allocator: the dam break went from 0 to 2 - 12 device mallocs per rollout.)
        if hint is not None:
            # Padded rows of row_stride(longest row of the previous step) entries: ONE candidate scan per query, no
            # count pass, no prefix scan, no host round trip (HBM is plentiful: 288 GB).  A row that outgrows the
            # stride is detected at the end of the step (one sync) and the step is repeated with the exact search.
            stride, total = row_stride(hint), self.totals.get(slot)
            if total is not None and queries.shape[0] * stride > 4 * total + (1 << 22):
                # ... unless the padded rows would be mostly air: a scene that has dissolved into spray (the 100k dam break
                # after ~40 steps: three times as many lattice points as at the start, most of them around single droplets,
                # next to a bulk whose rows still hold 2,700 entries) took 6 GB per list and 34 GB per step that way.  Then
                # count + scan + write into a buffer sized from the previous step's pairs -- still no host round trip.
                res = frs(points, queries, radius, hash_table=table, capacity_hint=total + total // 4)
            else:
                res = frs(points, queries, radius, hash_table=table, row_stride=stride, capacity_hint=self.caps.get(slot),
                          max_count=self._max_count_slot(points.device))
                self.caps[slot] = getattr(res, "capacity", None)
        else:
            res = frs(points, queries, radius, hash_table=table)
        self.pending.append((hkey, slot, res))
        self.lists[key] = res
        self.slot_of[key] = slot
        self._consumer(key, res)
        self.keepalive.append((points, queries))  # keep storage alive so data_ptr keys stay unique
        return res


class _PerThreadCache:
    """One _NeighborCache per thread: the virtual ranks of dmcf_amd.parallel.run_local_ranks are threads of one process and
    each must see its own step scope, search order and estimates (a real rank is a process and has one anyway)."""

    def __init__(self):
        object.__setattr__(self, "_tls", threading.local())

    def _get(self):
        c = getattr(self._tls, "cache", None)
        if c is None:
            c = self._tls.cache = _NeighborCache()
        return c

    def __getattr__(self, name):
        return getattr(self._get(), name)

    def __setattr__(self, name, value):
        setattr(self._get(), name, value)

    def __enter__(self):
        return self._get().__enter__()

    def __exit__(self, *exc):
        return self._get().__exit__(*exc)


_CACHE = _PerThreadCache()

class _CacheScope:
    """``with neighbor_cache(estimate=...)``: enters the process-wide cache; ``estimate`` (outermost scope only)
    turns on buffer sizes estimated from the previous step -- the caller must then be prepared to repeat the step
    on :class:`dmcf_amd.ops.NeighborCapacityExceeded` (Simulator.run_inference does)."""

    def __init__(self, estimate, key=None):
        self.estimate = estimate
        self.key = key

    def __enter__(self):
        if _CACHE.depth == 0:
            self.prev = _CACHE.use_hints
            _CACHE.use_hints = bool(self.estimate) and os.environ.get("DMCF_NO_ESTIMATE") != "1"
            _CACHE._get().select(self.key)
            self.outer = True
        else:
            self.outer = False
        return _CACHE.__enter__()

    def __exit__(self, *exc):
        try:
            return _CACHE.__exit__(*exc)
        finally:
            if self.outer:
                _CACHE.use_hints = self.prev


def neighbor_cache(estimate=False, key=None):
    """Context manager enabling per-step neighbour-list reuse (see :class:`_NeighborCache`).  ``key``: whose step this is
    -- e.g. (model, scene slot) -- so that the estimates one rollout leaves behind (row capacities, consumers per list, by
    position in the step's search sequence) are not applied to another model or another scene of the same batch."""
    return _CacheScope(estimate, key)


def pairs_ref(nns):
    """The list's pair count as a device scalar -- for the per-launch records of an installed ops.timer only.  Forming it is a
    reduction launch per list (ten per step of a 2-D model, 6 % of its dispatches); a step without a timer never asks: the layers'
    `_avg_neighbors` statistic (convolutions.py:385-388) resolves the list lazily, when somebody reads it."""
    return nns.total_ref if ops.timer is not None else None


def _scatter_guard(vals):
    if any(vals):
        raise RuntimeError("dmcf_cconv_scatter_forward dropped pairs: a block's box did not hold an output it reaches (plan / positions mismatch)")
    return False


def _hint_key(frs, points, queries, radius):
    """What a search is, for the estimates carried from one step to the next: radius, the size class of both point sets
    (half powers of two: particle counts drift by a few per cent per step, the lattices' with them) and the search's flags."""
    def size_class(n):
        return int(round(2.0 * math.log2(max(int(n), 1))))
    return (float(radius), size_class(points.shape[0]), size_class(queries.shape[0]), bool(frs.ignore_query_point),
            bool(frs.return_distances))


class _HintView:
    """The longest-row estimates as a sequence of numbers (tests / diagnostics); ``clear()`` forgets them."""

    def __init__(self, hints):
        self._hints = hints

    def __iter__(self):
        return iter(list(self._hints.values()))

    def __len__(self):
        return len(self._hints)

    def clear(self):
        self._hints.clear()


def neighbor_hints():
    """The longest-row estimates of the process-wide cache (tests / diagnostics)."""
    return _HintView(_CACHE.hints)


def row_stride(longest):
    """Row capacity for the padded single-pass search from the longest row of the previous step: 1/4 slack for long rows; short
    rows get MORE room -- twice the longest row up to 64 entries, + 64 beyond -- because there the maximum over a million rows is
    a noisy statistic (a splash that lands on a wall takes the longest s0 -> s0 row of the 100k dam break from ~35 to ~60 in one
    step: round 3's only repeated step of that 200-step rollout) and room is free: a row's unused tail is neither written nor
    read, and a 1.1M-row list of 80 instead of 56 entries is 110 MB more of 288 GB.  Rounded up to 1/8 of the enclosing power
    of two (a handful of distinct buffer sizes for the caching allocator)."""
    longest = int(longest)
    x = longest + max(longest // 4, min(longest, 64)) + 8
    g = max(8, 1 << max(x.bit_length() - 4, 0))  # 1/8 of the enclosing power of two
    return (x + g - 1) // g * g


def _init_tensor(name, shape, device):
    """Keras initializer strings used by the reference: 'uniform' = RandomUniform(-0.05, 0.05)
    (convolutions.py:156), 'zeros' (:157), 'glorot_uniform' (:169)."""
    if callable(name):
        return name(shape).to(device)
    if name == "uniform":
        return torch.empty(shape, device=device).uniform_(-0.05, 0.05)
    if name == "zeros":
        return torch.zeros(shape, device=device)
    if name == "glorot_uniform":
        fan_in, fan_out = shape[-2] * int(np.prod(shape[:-2])), shape[-1] * int(np.prod(shape[:-2]))
        limit = math.sqrt(6.0 / (fan_in + fan_out))
        return torch.empty(shape, device=device).uniform_(-limit, limit)
    raise NotImplementedError(f"initializer {name!r}")


_ACTIVATIONS = {None: None, "linear": None, "relu": torch.relu, "tanh": torch.tanh}


class PlainAttributes:
    """Mix-in for the modules of this package: attributes that are neither parameters, buffers nor sub-modules go straight to the
    instance dictionary.  torch.nn.Module.__setattr__ walks its registries and type checks for every assignment -- a layer sets
    a dozen bookkeeping attributes per call (the reference's ``.nns``, ``._conv_values``, ...), 300 assignments per step of a 2-D
    model: a fifth of that step's host time (tools/profile_small.py)."""

    def __setattr__(self, name, value):
        d = self.__dict__
        if (isinstance(value, (torch.nn.Parameter, torch.nn.Module)) or "_parameters" not in d or name in d["_parameters"]
                or name in d["_buffers"] or name in d["_modules"] or hasattr(type(self), name)):
            super().__setattr__(name, value)  # (registries; properties and other class-level descriptors keep their setters)
        else:
            d[name] = value


class ContinuousConv(PlainAttributes, torch.nn.Module):
    r"""Continuous convolution of Ummenhofer & Koltun (ICLR 2020) with DMCF's antisymmetric option:

        (f*g)(x) = 1/psi(x) * sum_{i in N(x,R)} a(x_i, x) f_i g(Lambda(x_i - x))

    Arguments are those of the reference layer (utils/convolutions.py:67-149).  Unsupported
    combinations raise instead of silently computing something else.
    """

    def __init__(self, filters, kernel_size, activation=None, use_bias=True, kernel_initializer="uniform",
                 bias_initializer="zeros", kernel_regularizer=None, bias_regularizer=None, align_corners=True,
                 coordinate_mapping="ball_to_cube_radial", interpolation="linear", normalize=True,
                 radius_search_ignore_query_points=False, radius_search_metric="L2", offset=None,
                 window_function=None, use_dense_layer_for_center=False,
                 dense_kernel_initializer="glorot_uniform", dense_kernel_regularizer=None, in_channels=None,
                 symmetric=False, sym_axis=2, circular=False, name=None, trainable=True, device=None, **kwargs):
        super().__init__()
        self.layer_name = name
        self.filters = filters
        self.kernel_size = [int(k) for k in kernel_size]
        if activation not in _ACTIVATIONS and not callable(activation):
            raise NotImplementedError(f"activation {activation!r}")
        self.activation = _ACTIVATIONS.get(activation, activation) if not callable(activation) else activation
        self.use_bias = use_bias
        self.kernel_initializer = kernel_initializer
        self.bias_initializer = bias_initializer
        self.align_corners = align_corners
        if coordinate_mapping not in ops.MAPPINGS:
            raise ValueError(f"coordinate_mapping {coordinate_mapping!r}")
        if interpolation not in ops.INTERPOLATIONS:
            raise ValueError(f"interpolation {interpolation!r}")
        self.coordinate_mapping = coordinate_mapping
        self.interpolation = interpolation
        self.normalize = normalize
        self.radius_search_ignore_query_points = radius_search_ignore_query_points
        self.radius_search_metric = radius_search_metric
        self.dense_kernel_initializer = dense_kernel_initializer
        self.symmetric = symmetric
        self.sym_axis = sym_axis
        self.circular = circular
        if offset is not None and any(float(o) != 0.0 for o in torch.as_tensor(offset).reshape(-1)):
            raise NotImplementedError("non-zero offset (never used by DMCF, convolutions.py:200-201)")
        self.offset = torch.zeros(3)
        self.window_function = window_function
        # convolutions.py:207-210
        self.fixed_radius_search = ops.FixedRadiusSearch(
            metric=self.radius_search_metric, ignore_query_point=self.radius_search_ignore_query_points,
            return_distances=self.window_function is not None)
        self.use_dense_layer_for_center = use_dense_layer_for_center
        self.dense = None
        self.in_channels = None
        self.kernel = None
        self.bias = None
        self._device = device
        self.nns = None
        self._direct_kernel = None  # is this layer served by cconv_direct_kernel (learnt at its first call in a step)
        # what the MODEL knows about this layer's rows from its configuration (include/dmcf_hip.h, row_length_hint): 0 unknown,
        # 1 = the network's base radius (tens of neighbours), 2 = a wider radius (hundreds); models/hrnet.py sets it
        self.row_length_hint = 0
        self.accumulate_into = None  # one-shot, see forward
        self.extra_bias = None
        self._packed = {}  # the workspace of the last dmcf_cconv_forward and what its packed filter was made from (ops.cconv_forward)

    def invalidate_packed(self):
        """Drops the cached packed filter (ops.cconv_forward).  Needed only after writing the weights through ``.data`` --
        the one way to change a tensor that its version counter does not see."""
        self._packed.clear()

    def _load_from_state_dict(self, *args, **kwargs):
        self._packed.clear()
        return super()._load_from_state_dict(*args, **kwargs)

    # -- weights (lazy, from the first input's channel count: convolutions.py:228-275) ----------------
    def build(self, in_channels, device=None):
        self._packed.clear()
        device = device or self._device or "cuda"
        self.in_channels = int(in_channels)
        if self.circular:
            kshape = (math.ceil(max(self.kernel_size) / 2), self.in_channels, self.filters)  # :231-234
        elif self.symmetric:
            sh = list(self.kernel_size)
            assert sh[self.sym_axis] % 2 == 0  # :244
            sh[self.sym_axis] = sh[self.sym_axis] // 2
            kshape = (*sh, self.in_channels, self.filters)
        else:
            kshape = (*self.kernel_size, self.in_channels, self.filters)
        self.kernel = torch.nn.Parameter(_init_tensor(self.kernel_initializer, kshape, device), requires_grad=False)
        if self.use_bias:
            self.bias = torch.nn.Parameter(_init_tensor(self.bias_initializer, (self.filters,), device),
                                           requires_grad=False)
        if self.use_dense_layer_for_center:
            w = _init_tensor(self.dense_kernel_initializer, (self.in_channels, self.filters), device)
            self.dense = torch.nn.Parameter(w, requires_grad=False)

    def _expanded_kernel(self):
        """circular kernels (convolutions.py:395-409): ring index = max_axis floor(|grid offset|)."""
        ks = self.kernel_size
        zr, yr, xr = torch.meshgrid(*[torch.arange(k, device=self.kernel.device) for k in ks], indexing="ij")
        size_xyz = torch.tensor(ks[::-1], dtype=torch.float32, device=self.kernel.device)
        gp = torch.stack([xr, yr, zr], dim=-1).to(torch.float32) - size_xyz / 2.0 + 0.5
        mask = (gp * 2.0) / size_xyz
        idx = torch.floor(gp.abs()).max(dim=-1).values.to(torch.int64)
        kernel = self.kernel[idx]
        if self.symmetric:
            # NOTE: the reference multiplies by mask[..., None, :] (shape [D,H,W,1,3]), which only
            # broadcasts when filters == 3 (convolutions.py:408-409)
            kernel = kernel * mask.unsqueeze(-2)
        return kernel

    @torch.no_grad()
    def forward(self, inp_features, inp_positions, out_positions, extents, inp_importance=None,
                fixed_radius_search_hash_table=None, user_neighbors_index=None, user_neighbors_row_splits=None,
                user_neighbors_importance=None):
        if self.kernel is None:
            self.build(inp_features.shape[-1], inp_features.device)
        # one-shot requests of the caller (models/hrnet.py): add the result to this tensor in the kernel's epilogue
        # (DMCF_FLAG_ACCUMULATE) instead of returning a new one, and add this vector to the layer's bias
        acc, extra_bias = self.accumulate_into, self.extra_bias
        d = self.__dict__  # (bookkeeping attributes written straight to the instance dictionary: PlainAttributes would put them there too)
        d["accumulate_into"] = d["extra_bias"] = None
        if acc is not None and (self.use_dense_layer_for_center or self.activation is not None or not acc.is_contiguous()
                                or tuple(acc.shape) != (out_positions.shape[0], self.filters) or acc.dtype != torch.float32):
            # not expressible in the epilogue: the plain call, then the sums (the extra bias OUTSIDE the layer's activation)
            acc.add_(self.forward(inp_features, inp_positions, out_positions, extents, inp_importance,
                                  fixed_radius_search_hash_table, user_neighbors_index, user_neighbors_row_splits,
                                  user_neighbors_importance))
            return acc if extra_bias is None else acc.add_(extra_bias)
        if isinstance(extents, torch.Tensor):
            if extents.dim() > 0 and extents.numel() != 1:
                raise NotImplementedError("per-point extents (RadiusSearch, convolutions.py:366-370) are never "
                                          "used by DMCF and not implemented")
            extent = float(extents)
        else:
            extent = float(np.float32(extents))
        window, window_fac, neighbors_value, n_pairs_ref, row_count = None, 1.0, None, None, None
        skip_self = False
        if user_neighbors_index is not None and user_neighbors_row_splits is not None:  # :341-349
            neighbors_index, neighbors_row_splits = user_neighbors_index, user_neighbors_row_splits
            if user_neighbors_importance is not None and user_neighbors_importance.numel() > 0:
                window, neighbors_value = "explicit", user_neighbors_importance
        else:
            lat = self._lattice_form(inp_features, inp_positions, out_positions, inp_importance,
                                     fixed_radius_search_hash_table, extent)
            if lat is not None:
                # both point sets are grid_pos lattices of this step: no search, no per-pair geometry
                # (dmcf_lattice_conv_forward; DMCF_LATTICE_CONV=0 keeps the neighbour-list form)
                d["nns"] = None
                fuse_bias = self.use_bias and not self.use_dense_layer_for_center
                d["_n_out_last"] = out_positions.shape[0]
                d["_pairs_last"] = 0  # no pair list in this form
                out_features = lat.conv(
                    ops, self.kernel, inp_features, out_positions.shape[0], extent,
                    window=self.window_function.name, window_fac=self.window_function.fac,
                    align_corners=self.align_corners, coordinate_mapping=self.coordinate_mapping,
                    interpolation=self.interpolation, bias=self._epilogue_bias(fuse_bias, extra_bias),
                    out=acc, accumulate=acc is not None)
                stray = lat.strays()
                if stray is not None:
                    # output points outside the lattice's core (stray particles' cells, dmcf_amd/lattice.py): the same layer
                    # through a neighbour list over ALL input points, added to their rows of the result (rows the stencil form
                    # does not write: zero, or the value to accumulate to).  The row set has an estimated capacity inside a
                    # rollout: ``valid`` zeroes the padding rows
                    idx_s, pos_s, valid = stray
                    radius = float(np.float32(0.5) * np.float32(extent))
                    nns = _CACHE.search(self.fixed_radius_search, inp_positions, pos_s, radius, distances=False)
                    ni, nrs, raw_dist = nns.raw()
                    res = ops.cconv_forward(
                        self.kernel, pos_s, extent, inp_positions, inp_features, ni, nrs, neighbors_value=raw_dist,
                        window=self.window_function.name, window_fac=self.window_function.fac, align_corners=self.align_corners,
                        coordinate_mapping=self.coordinate_mapping, interpolation=self.interpolation,
                        bias=self._epilogue_bias(fuse_bias, extra_bias), n_pairs_ref=pairs_ref(nns),
                        neighbors_row_count=getattr(nns, "row_count", None), row_length_hint=1)
                    if valid is not None:
                        res = res * valid
                    out_features.index_add_(0, idx_s, res)
                d["_conv_values"], d["_conv_output"] = None, (None if _CACHE.depth > 0 else out_features)
                return self._finish(out_features, inp_features, extra_bias if self.use_dense_layer_for_center else None)
            radius = float(np.float32(0.5) * np.float32(extent))  # :353
            sct = self._scatter_form(inp_features, inp_positions, out_positions, inp_importance, fixed_radius_search_hash_table,
                                     radius)
            if sct is not None:
                # particles -> coarse lattice with 4 output channels: filter first, input stationary, over the TRANSPOSED list
                # (dmcf_cconv_scatter_forward; DMCF_SCATTER_CONV=0 keeps the gather form) -- the list of the later
                # lattice -> particles layers of the step, searched here if this layer is the first to ask for it
                voxel, m = sct
                tl = _CACHE.search(self.fixed_radius_search, out_positions, inp_positions, radius, distances=False)
                t_idx, t_rb, _ = tl.raw()
                d["nns"] = None
                d["_n_out_last"] = out_positions.shape[0]
                d["_pairs_last"] = tl
                fuse_bias = self.use_bias and not self.use_dense_layer_for_center
                # (the kernel's guard -- a pair outside its block's box, i.e. a plan that does not match the positions -- is read with
                # the step's one synchronisation and fails loudly; it has never fired)
                flag = _CACHE._max_count_slot(inp_positions.device) if _CACHE.depth > 0 else torch.zeros(1, dtype=torch.int32, device=inp_positions.device)
                out_features = ops.cconv_scatter_forward(
                    self.kernel, out_positions, extent, inp_positions, inp_features, t_idx, t_rb, getattr(tl, "row_count", None),
                    _CACHE.scatter_plan(inp_positions, out_positions, voxel, radius, m), window=self.window_function.name,
                    window_fac=self.window_function.fac, bias=self._epilogue_bias(fuse_bias, extra_bias), out=acc,
                    accumulate=acc is not None, n_pairs_ref=pairs_ref(tl), error_flag=flag)
                _CACHE.report(flag, _scatter_guard)
                d["_conv_values"], d["_conv_output"] = None, (None if _CACHE.depth > 0 else out_features)
                return self._finish(out_features, inp_features, extra_bias if self.use_dense_layer_for_center else None)
            if fixed_radius_search_hash_table is not None:
                d["nns"] = self.fixed_radius_search(inp_positions, out_positions, radius,
                                                    hash_table=fixed_radius_search_hash_table)
            else:
                shared = None
                if self._shares_list():
                    shared = _CACHE.with_query_points(self.fixed_radius_search, inp_positions, out_positions, radius)
                skip_self = shared is not None
                d["nns"] = shared if skip_self else _CACHE.search(
                    self.fixed_radius_search, inp_positions, out_positions, radius,
                    distances=not isinstance(self.window_function, WindowFunction))
            # raw(): buffers that may be longer than P (no host round trip); the kernels only follow row_splits
            neighbors_index, neighbors_row_splits, raw_dist = self.nns.raw()
            row_count = getattr(self.nns, "row_count", None)  # padded rows of the single-pass search
            n_pairs_ref = pairs_ref(self.nns)
            if self.window_function is not None:  # :359-379
                if isinstance(self.window_function, WindowFunction):
                    window, window_fac = self.window_function.name, self.window_function.fac
                    neighbors_value = raw_dist  # d^2; q = d^2/R^2 is formed in the kernel
                else:
                    q = self.nns.neighbors_distance / (np.float32(radius) * np.float32(radius))
                    neighbors_index = self.nns.neighbors_index  # compact CSR form (synchronises)
                    if row_count is not None:
                        neighbors_row_splits, row_count = self.nns.csr_row_splits, None
                    window, neighbors_value = "explicit", self.window_function(q).to(torch.float32)
        # stats (convolutions.py:385-388) are formed lazily (property _avg_neighbors): no host sync here
        d["_n_out_last"] = out_positions.shape[0]
        d["_pairs_last"] = self.nns if (n_pairs_ref is None and getattr(self, "nns", None) is not None and user_neighbors_index is None) \
            else (n_pairs_ref if n_pairs_ref is not None else neighbors_index.shape[0])

        kernel = self.kernel
        symmetric = self.symmetric
        if self.circular:
            kernel = self._expanded_kernel()
            symmetric = False  # the mask already made it antisymmetric; second pass still applies below
            if self.symmetric:
                raise NotImplementedError("circular + symmetric kernels (filters must be 3 in the reference; "
                                          "no shipped config uses circular: True)")
        # The reference keeps the operands of the last call for inspection (convolutions.py:398-413).  Inside a rollout
        # step (neighbor_cache scope) that would pin every layer's neighbour list (GBs each) and activations into the
        # next step: there only the small entries are kept.
        in_step = _CACHE.depth > 0
        d["_conv_values"] = {
            "filters": kernel, "out_positions": out_positions, "extents": extent, "offset": self.offset,
            "inp_positions": inp_positions, "inp_features": None if in_step else inp_features,
            "inp_importance": inp_importance,
            "neighbors_index": None if in_step else neighbors_index, "neighbors_row_splits": neighbors_row_splits,
            "neighbors_importance": None if in_step else neighbors_value, "align_corners": self.align_corners,
            "coordinate_mapping": self.coordinate_mapping, "interpolation": self.interpolation,
            "normalize": self.normalize,
        }
        if symmetric and self.normalize:
            raise NotImplementedError("symmetric=True with normalize=True (DMCF always uses normalize=False, "
                                      "models/pbf_model.py:203)")
        fuse_bias = self.use_bias and not self.use_dense_layer_for_center
        out_features = ops.cconv_forward(
            kernel, out_positions, extent, inp_positions, inp_features, neighbors_index, neighbors_row_splits,
            neighbors_value=neighbors_value, window=window, window_fac=window_fac, inp_importance=inp_importance,
            align_corners=self.align_corners, coordinate_mapping=self.coordinate_mapping,
            interpolation=self.interpolation, normalize=self.normalize, symmetric=symmetric, sym_axis=self.sym_axis,
            bias=self._epilogue_bias(fuse_bias, extra_bias), n_pairs_ref=n_pairs_ref,
            neighbors_row_count=row_count, skip_self=skip_self, row_length_hint=self.row_length_hint,
            out=acc, accumulate=acc is not None,
            packed_cache=self._packed if (kernel is self.kernel and os.environ.get("DMCF_CACHE_PACKED_FILTERS", "1") != "0")
            else None)  # (a derived filter tensor -- circular -- is a new object every call: never cached)
        if self._direct_kernel is None and in_step and self.radius_search_ignore_query_points:
            # (asked once per layer: the dispatch looks at the layer, never at the list)
            self._direct_kernel = ops.cconv_forward(
                kernel, out_positions, extent, inp_positions, inp_features, neighbors_index, neighbors_row_splits,
                neighbors_value=neighbors_value, window=window, window_fac=window_fac, inp_importance=inp_importance,
                align_corners=self.align_corners, coordinate_mapping=self.coordinate_mapping,
                interpolation=self.interpolation, normalize=self.normalize, symmetric=symmetric, sym_axis=self.sym_axis,
                neighbors_row_count=row_count, name_only=True).startswith("cconv_direct_kernel")
            if self._shares_list() and user_neighbors_index is None and fixed_radius_search_hash_table is None:
                # announce this layer as one more consumer of the list it will take from the next step on (the cache hands a
                # list's buffers back after the number of consumers it saw in the previous step)
                _CACHE.with_query_points(self.fixed_radius_search, inp_positions, out_positions,
                                         float(np.float32(0.5) * np.float32(extent)))
        d["_conv_output"] = None if in_step else out_features
        return self._finish(out_features, inp_features, extra_bias if self.use_dense_layer_for_center else None)

    def _epilogue_bias(self, fuse_bias, extra_bias):
        """The vector the kernel's epilogue adds: the layer's bias (when the epilogue may form it) + the caller's."""
        if self.use_dense_layer_for_center:
            return None  # (bias and the caller's vector follow the dense term, _finish)
        bias = self.bias if fuse_bias else None
        if extra_bias is None:
            return bias
        return extra_bias if bias is None else bias + extra_bias

    def _shares_list(self):
        """May this layer take the list of the same search WITH the query points (see _NeighborCache.with_query_points)?  Only
        a layer that ignores them, evaluates a named window in the kernel and is served by the kernel that implements
        DMCF_FLAG_SKIP_SELF; DMCF_SHARE_LISTS=0 turns it off."""
        return (self.radius_search_ignore_query_points and self._direct_kernel is True and self.radius_search_metric == "L2"
                and isinstance(self.window_function, WindowFunction) and os.environ.get("DMCF_SHARE_LISTS", "1") != "0")

    def _finish(self, out_features, inp_features, extra_bias=None):
        if self.use_dense_layer_for_center:  # :462-464
            dense_output = inp_features @ self.dense
            self._dense_output = None if _CACHE.depth > 0 else dense_output
            out_features = out_features + dense_output
            if self.use_bias:
                out_features = out_features + self.bias
        if extra_bias is not None:
            out_features = out_features + extra_bias
        if self.activation is not None:
            out_features = self.activation(out_features)
        return out_features

    def _scatter_form(self, inp_features, inp_positions, out_positions, inp_importance, hash_table, radius):
        """(voxel, block_cells) if dmcf_cconv_scatter_forward applies to this call: the outputs are a registered grid_pos lattice
        with one spacing on all axes, the inputs are not a lattice of that family (then the stencil form applies), few output
        channels, a 4x4x4 filter with the options that kernel implements, and the default (symmetric) neighbour set -- the
        transposed list then holds the same pairs as the forward one."""
        from .. import lattice
        if (os.environ.get("DMCF_SCATTER_CONV", "1") == "0" or not lattice.enabled() or hash_table is not None
                or inp_importance is not None or self.symmetric or self.circular or self.normalize
                or not isinstance(self.window_function, WindowFunction) or self.window_function.name != "poly6"
                or self.radius_search_ignore_query_points or self.radius_search_metric != "L2" or not inp_features.is_cuda
                or self.kernel_size != [4, 4, 4] or self.in_channels > 32 or self.filters not in SCATTER_OUT_CHANNELS
                or not self.align_corners or self.coordinate_mapping != "ball_to_cube_volume_preserving"
                or self.interpolation != "linear" or inp_positions.shape[0] < SCATTER_MIN_INPUTS):
            return None
        info = lattice.lookup(out_positions)
        if info is None or not (info.voxel[0] == info.voxel[1] == info.voxel[2]) or max(info.dims) > SCATTER_MAX_LATTICE_CELLS:
            return None
        reach = ops.scatter_reach(radius, info.voxel[0])
        m = min(ops.SCATTER_BLOCK_CELLS, (13 if self.filters == 4 else 11) - 1 - 2 * reach)
        if m < 1 or out_positions.shape[0] * 4 > inp_positions.shape[0]:
            return None  # (a fine lattice: few pairs per flushed slot, the gather form is faster -- DESIGN.md section 4.2, splat S)
        return info.voxel[0], m

    def _lattice_form(self, inp_features, inp_positions, out_positions, inp_importance, hash_table, extent):
        """The LatticePair for this call if dmcf_lattice_conv_forward applies: both position tensors registered grid_pos
        lattices of one family (dmcf_amd/lattice.py), a named window, none of the options that form needs a pair list for."""
        from .. import lattice
        if (hash_table is not None or inp_importance is not None or self.symmetric or self.circular or self.normalize
                or not isinstance(self.window_function, WindowFunction) or self.radius_search_ignore_query_points
                or self.radius_search_metric != "L2" or not inp_features.is_cuda
                or self.in_channels not in (4, 8) or self.filters > 32):
            return None
        # (None too when the scene reaches so far from the origin that the nominal offsets d * voxel of this form and the
        # reference's differences of rounded positions part by more than the parity bar allows: lattice.MAX_X_OVER_EXTENT)
        lp = lattice.pair(inp_positions, out_positions, extent)
        if lp is not None:
            # a bounding box blown up by stray particles (a splash, a particle that left the scene): the dense zero-filled
            # volume would be mostly empty, or not fit the kernel's 32-bit addressing -- the neighbour-list form does not care
            cells = lp.volume_cells(ops, extent, inp_features.device)
            if (cells * self.in_channels > lattice.MAX_VOLUME_FLOATS
                    or cells > max(1 << 20, lattice.MAX_CELLS_PER_POINT * inp_positions.shape[0])):
                return None
        return lp

    call = forward

    @property
    def _avg_neighbors(self):
        """pairs / outputs of the last call (convolutions.py:385-388); synchronises when read."""
        pairs = self._pairs_last
        if hasattr(pairs, "total_ref"):  # (the list itself: its total is formed now, not in every step)
            pairs = pairs.total_ref
        pairs = int(pairs.item()) if isinstance(pairs, torch.Tensor) else pairs
        return pairs / max(self._n_out_last, 1)

    def compute_output_shape(self, inp_features_shape):
        return (None, self.filters)


class PointSampling(PlainAttributes, torch.nn.Module):
    """Mirror of the reference's ``PointSampling`` (utils/convolutions.py:888-1061): resamples features from one
    point set to another -- a ContinuousConv with a fixed 1x1x1 identity filter, optional window and
    normalisation by the summed window values.  Used by ``dens_norm`` to carry the density to the coarse
    point sets (models/pbf_model.py:177-181, 421-431)."""

    def __init__(self, window_function=None, normalize=True, name=None, **kwargs):
        super().__init__()
        self.normalize = normalize
        self.window_function = window_function
        self.fixed_radius_search = ops.FixedRadiusSearch(metric="L2", ignore_query_point=False,
                                                         return_distances=window_function is not None)
        self.kernel = None
        self.nns = None
        self.layer_name = name

    def build(self, in_channels, device=None):  # :925-929
        self.in_channels = in_channels
        self.kernel = torch.eye(in_channels, dtype=torch.float32, device=device).reshape(1, 1, 1, in_channels, in_channels)

    def forward(self, inp_features, inp_positions, out_positions, extents, inp_importance=None,
                fixed_radius_search_hash_table=None, user_neighbors_index=None, user_neighbors_row_splits=None,
                user_neighbors_importance=None):
        if self.kernel is None or self.kernel.shape[-1] != inp_features.shape[-1] or self.kernel.device != inp_features.device:
            self.build(inp_features.shape[-1], inp_features.device)
        if isinstance(extents, torch.Tensor) and extents.dim() > 0 and extents.numel() != 1:
            raise NotImplementedError("per-point extents (RadiusSearch, convolutions.py:1006-1010) are not implemented")
        extent = float(np.float32(float(extents)))
        window, window_fac, neighbors_value, n_pairs_ref, row_count = None, 1.0, None, None, None
        if user_neighbors_index is not None and user_neighbors_row_splits is not None:  # :984-993
            neighbors_index, neighbors_row_splits = user_neighbors_index, user_neighbors_row_splits
            if user_neighbors_importance is not None and user_neighbors_importance.numel() > 0:
                window, neighbors_value = "explicit", user_neighbors_importance
        else:
            radius = float(np.float32(0.5) * np.float32(extent))  # :997
            if fixed_radius_search_hash_table is not None:
                self.nns = self.fixed_radius_search(inp_positions, out_positions, radius,
                                                    hash_table=fixed_radius_search_hash_table)
            else:
                self.nns = _CACHE.search(self.fixed_radius_search, inp_positions, out_positions, radius,
                                         distances=not isinstance(self.window_function, WindowFunction))
            neighbors_index, neighbors_row_splits, raw_dist = self.nns.raw()
            row_count = getattr(self.nns, "row_count", None)
            n_pairs_ref = pairs_ref(self.nns)
            if self.window_function is not None:  # :1015-1019
                if isinstance(self.window_function, WindowFunction):
                    window, window_fac, neighbors_value = self.window_function.name, self.window_function.fac, raw_dist
                else:
                    q = self.nns.neighbors_distance / (np.float32(radius) * np.float32(radius))
                    neighbors_index = self.nns.neighbors_index
                    if row_count is not None:
                        neighbors_row_splits, row_count = self.nns.csr_row_splits, None
                    window, neighbors_value = "explicit", self.window_function(q).to(torch.float32)
        self._n_out_last = out_positions.shape[0]
        self._pairs_last = n_pairs_ref if n_pairs_ref is not None else (self.nns if user_neighbors_index is None else neighbors_index.shape[0])
        # ml3d.ops.continuous_conv with its defaults (:1038-1052): align_corners=False, ball_to_cube_radial, linear --
        # irrelevant for a one-cell filter, every neighbour puts all its weight on that cell
        out = ops.cconv_forward(self.kernel, out_positions, extent, inp_positions, inp_features, neighbors_index,
                                neighbors_row_splits, neighbors_value=neighbors_value, window=window, window_fac=window_fac,
                                inp_importance=inp_importance, align_corners=False, coordinate_mapping="ball_to_cube_radial",
                                interpolation="linear", normalize=self.normalize, n_pairs_ref=n_pairs_ref,
                                neighbors_row_count=row_count)
        self._conv_output = None if _CACHE.depth > 0 else out  # see ContinuousConv.forward
        return out

    call = forward
