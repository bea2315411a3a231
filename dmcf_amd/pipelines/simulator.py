"""Simulator -- mirror of the inference surface of the reference's ``pipelines/simulator.py:37-109``.

``run_inference(inputs)`` (simulator.py:57-71) takes a list of per-scene input lists
``[pos, vel, acc|None, feats|None, box, box_normals]`` and returns ``[pos', vel'] + inputs[2:]`` per scene so
the result is fed back verbatim; ``run_rollout(inputs, timesteps)`` (:73-109) loops it.  The reference has
no ``step()``; it is provided as an alias because BASELINE.json names that surface.

Training / validation / HDF5 writing are out of scope of the hot path.  The number of particles may change
between steps (run_sample.py:173-177 adds inflow), so nothing here assumes a fixed N.
"""
import logging
import os
import time

import numpy as np
import torch

from ..utils.config import Config
from .. import ops
from ..utils.convolutions import neighbor_cache

log = logging.getLogger(__name__)


# 1M fluid + 0.12M boundary particles peak at 16 GB of live buffers and ~45 GB of pool once the lists have grown (DESIGN.md
# section 4.1): 40 KiB per point
RESERVE_BYTES_PER_POINT = 40 * 1024


def reserve_for_scene(reserve_gib, n_points, device):
    """The ``reserve_gib`` rule of Simulator / ShardedSimulator for a scene of ``n_points`` particles (fluid + boundary): "auto" =
    RESERVE_BYTES_PER_POINT each, at most an eighth of the device, at least 0.25 GiB.  Returns what ops.reserve_device_memory reports (0.0 when nothing was asked for)."""
    gib = reserve_gib
    if gib and device is not None and torch.device(device).type == "cuda":
        # torch's allocator serves requests up to 1 MB from 2 MB segments of their own ("small pool"): row splits, counts,
        # headers, the 2-D scenes' whole lists.  That pool grows one hipMalloc at a time whenever a step needs one block more
        # than any step before it (round 3: six such steps in the 3200-step rollout) -- hold 64 MB of it from the start.
        small = [torch.empty(1 << 20, dtype=torch.uint8, device=device) for _ in range(64)]
        del small
        # ... and load the code of the library kernels a rollout may meet late: grid_pos's sort-based form (torch.unique +
        # argsort, taken once stray particles make the lattices' bounding box too sparse for a dense cell table) cost the
        # 100k dam break 300 ms in the step that first needed it
        from ..utils.tools.losses import _unique_first_occurrence
        _unique_first_occurrence(torch.arange(8, dtype=torch.int64, device=device) % 3)
    if gib == "auto":
        gib = n_points * RESERVE_BYTES_PER_POINT / 2 ** 30
        gib = min(gib, torch.cuda.mem_get_info(device)[1] / 2 ** 30 / 8)  # (total device memory)
        gib = max(gib, 0.25)  # (the 2-D scenes: their lists are a few MB, but grow with every particle that leaves the tank)
        log.info("reserve_gib='auto': %.2f GiB of device memory go to the caching allocator's pool before the first step", gib)
    return ops.reserve_device_memory(float(gib), device) if gib else 0.0


class steady_steps:
    """``with steady_steps(): for ...: sim.step(...)`` -- Python's cyclic garbage collector off for the duration of a rollout
    loop (the young generation is collected by hand every ``every`` steps).  A step of the 2-D models is ~5 ms of host work
    (40+ launches per layer stack); a generation-2 collection in the middle of one is what the rollouts' p99 showed
    (WBC-SPH, 3200 steps: p99 / median 1.2 - 1.55 with the collector running, 1.22 without, tools/long_rollout.py).
    ``Simulator.run_rollout`` runs inside one.  The switch is process wide (other threads see the collector off as well) for
    the duration of the ``with`` block."""

    def __init__(self, every=256):
        self.every, self.n = int(every), 0

    def __enter__(self):
        import gc
        self.gc, self.was = gc, gc.isenabled()
        gc.collect()
        gc.disable()
        return self

    def tick(self, full=False):
        """Once per step.  ``full``: a whole collection now (a step was repeated: the exception's traceback may hold device
        tensors in a cycle); otherwise the young generation every ``every`` steps and everything every 16 x ``every`` -- cycles
        that hold device buffers (closures of the neighbour lists, tracebacks) must not live until the rollout ends."""
        self.n += 1
        if full or self.n % (16 * self.every) == 0:
            self.gc.collect()
        elif self.n % self.every == 0:
            self.gc.collect(0)

    def __exit__(self, *exc):
        if self.was:
            self.gc.enable()
        return False


class Simulator:
    def __init__(self, model, dataset=None, name="Simulator", main_log_dir="./logs/", device="cuda", split="train",
                 reserve_gib=None, **kwargs):
        """``reserve_gib`` (not in the reference; a ``pipeline:`` key like the others, OFF unless asked for): device memory handed
        to torch's caching allocator as one block before the first step (ops.reserve_device_memory), so that the multi-GB
        neighbour-list buffers of a large scene -- and the bigger ones it grows into -- never wait for a hipMalloc in the middle
        of a rollout.  "auto" = RESERVE_BYTES_PER_POINT per particle (fluid + boundary) of the first scene, at most an EIGHTH of
        the device (43 GiB would be the rule's figure for 1M particles: 36 GiB on a 288 GB part), logged when taken; a number =
        that many GiB; 0 / None (the default) = none -- the library itself never allocates device memory (INTEGRATION.md
        section 4), and a caller who did not ask does not lose a slice of the device to this module either.  bench.py asks for
        "auto" and reports what it got (``scene_state.reserved_gib``)."""
        self.cfg = Config(dict(kwargs, name=name, main_log_dir=main_log_dir, device=device, split=split, reserve_gib=reserve_gib))
        self.name = name
        self.model = model
        self.dataset = dataset
        if device in ("gpu", "cuda"):
            device = "cuda"
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the DMCF hot path runs on the GPU only (no CPU fallback)")
        self.timing = []
        self.reserve_gib = reserve_gib
        self.reserved_gib = None  # what the first step took from the device (None: not asked yet)
        self._slot0 = 0  # scene slot of inputs[0] (run_rollout feeds its scenes one at a time)
        self.repeated_steps = 0  # steps repeated with exact buffer sizes after a NeighborCapacityExceeded
        # base_pipeline.py:46-63: <main_log_dir | output_dir>/<Model>_<dataset>_<version>
        tag = "_".join([type(model).__name__, dataset.name if dataset is not None and hasattr(dataset, "name") else "",
                        str(kwargs.get("version", ""))])
        self.cfg.logs_dir = os.path.join(main_log_dir, tag)
        self.cfg.out_dir = os.path.join(kwargs.get("output_dir", "./output"), tag)

    def _to_device(self, x):
        if x is None:
            return None
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        return x.to(self.device, dtype=torch.float32)

    def _reserve(self, inputs):
        """Once, before the first step: see ``reserve_gib`` in __init__."""
        n = max((int(s[0].shape[0]) + (int(s[4].shape[0]) if s[4] is not None else 0) for s in inputs), default=0)
        self.reserved_gib = reserve_for_scene(self.reserve_gib, n, self.device)

    @torch.no_grad()
    def run_inference(self, inputs):
        """simulator.py:57-71."""
        if self.reserved_gib is None:
            self._reserve(inputs)
        results = []
        for bi in range(len(inputs)):
            try:
                with neighbor_cache(estimate=True, key=(id(self.model), bi + self._slot0)):
                    pos, vel = self.model(inputs[bi], training=False)
            except ops.NeighborCapacityExceeded:
                # a neighbour list grew by more than the slack since the previous step: repeat with exact sizes
                self.repeated_steps += 1
                with neighbor_cache(estimate=False, key=(id(self.model), bi + self._slot0)):
                    pos, vel = self.model(inputs[bi], training=False)
            results.append([pos, vel] + list(inputs[bi][2:]))
        return results

    def step(self, inputs):
        """Alias of :meth:`run_inference` (one simulated time step)."""
        return self.run_inference(inputs)

    @torch.no_grad()
    def run_rollout(self, inputs, timesteps=2):
        """simulator.py:73-109.  ``inputs``: list of dicts with [T,N,3] arrays ``pos, vel, grav, box,
        box_normals`` (frame 0 is used), as produced by get_rollout (dataset_reader_physics.py:410-456)."""
        inputs = [[self._to_device(data["pos"][0]), self._to_device(data["vel"][0]),
                   self._to_device(data["grav"][0]) if data.get("grav") is not None and data["grav"][0] is not None
                   else None, None, self._to_device(data["box"][0]), self._to_device(data["box_normals"][0])]
                  for data in inputs]
        results = [[] for _ in range(len(inputs))]
        self.run_inference(inputs[:1])  # "dummy init": builds the lazily created weights (simulator.py:94)
        timing = []
        for i in range(len(inputs)):
            results[i].append(inputs[i])
        with steady_steps() as steady:
            for _ in range(timesteps - 1):
                torch.cuda.synchronize(self.device)
                start = time.time()
                repeated = self.repeated_steps
                for i in range(len(inputs)):
                    self._slot0 = i  # each scene keeps its own buffer-size estimates
                    inputs[i] = self.run_inference(inputs[i:i + 1])[0]
                self._slot0 = 0
                torch.cuda.synchronize(self.device)
                timing.append(time.time() - start)
                for i in range(len(inputs)):
                    results[i].append(inputs[i])
                steady.tick(full=self.repeated_steps != repeated)
        self.timing = timing
        if timing:
            log.info("Average runtime: %.05f" % (np.mean(timing) / len(inputs)))
        return results

    def load_ckpt(self, ckpt_path):
        """base_pipeline.py:155-187 for inference: read a TensorFlow tensor-bundle checkpoint (``<dir>/ckpt`` prefix, or
        the newest ``ckpt-<n>`` in a directory) into the model; returns the epoch (0 without a checkpoint: the model then
        runs on its initialisers, as the reference does)."""
        from ..utils import tf_checkpoint as tc
        if not ckpt_path:
            log.info("No checkpoint")
            return 0
        prefix, epoch = ckpt_path, 0  # an explicit checkpoint prefix: epoch 0 (base_pipeline.py:171-175)
        if os.path.isdir(ckpt_path):
            import glob
            import re
            idx = sorted(glob.glob(os.path.join(ckpt_path, "*.index")),
                         key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))] or [0])
            if not idx:
                log.info("No checkpoint")
                return 0
            prefix = idx[-1][:-len(".index")]
            # the newest checkpoint of a directory (manager.latest_checkpoint): 'ckpt-<n>' was written at the end of epoch
            # (n - 1) * save_ckpt_freq, the run continues with the next one (base_pipeline.py:176-185)
            epoch = tc.checkpoint_epoch(prefix, int(self.cfg.get("save_ckpt_freq", 1) or 1))
        log.info("Loading checkpoint %s", prefix)
        tc.load_into_model(self.model, tc.load_checkpoint(prefix), device=self.device)
        return epoch

    def run_test(self, epoch=None):
        """simulator.py:111-165: roll out every scene of the test split over its full length and write
        ``<out_dir>/visual/<scene>/<epoch>.hdf5`` with the datasets pred / gt / bnd (an HDF5 file utils/draw_sim2d.py reads:
        through h5py when installed, else by the built-in writer).  Returns the list of output paths."""
        from ..datasets import get_rollout, write_results
        cfg = self.cfg
        gen = dict(cfg.get("data_generator") or {})
        test_kw = dict(gen.pop("test", None) or {})
        for k in ("train", "valid"):
            gen.pop(k, None)
        test_data = get_rollout(self.dataset.test, **gen, **test_kw)
        if epoch is None:
            epoch = self.load_ckpt(self.model.cfg.get("ckpt_path"))
        log.info("Started testing")
        results = self.run_rollout(test_data, test_data[0]["pos"].shape[0])
        paths = []
        for i in range(len(results)):
            data = test_data[i]
            pos = np.stack([r[0].cpu().numpy() for r in results[i]])
            out_dir = os.path.join(cfg.out_dir, "visual", "%04d" % i)
            os.makedirs(out_dir, exist_ok=True)
            output = [(pos, {"name": "pred", "type": "PARTICLE"}), (data["pos"], {"name": "gt", "type": "PARTICLE"}),
                      (data["box"][0], {"name": "bnd", "type": "PARTICLE"})]
            path = os.path.join(out_dir, "%04d.hdf5" % epoch)
            write_results(path, self.model.name, output)
            # simulator.py:155-162: write first, THEN drop the scene directory's other result files (a failed write must not
            # cost the previous results)
            for stale in os.listdir(out_dir):
                if stale.endswith((".hdf5", ".npz")) and os.path.join(out_dir, stale) != path:
                    os.remove(os.path.join(out_dir, stale))
            paths.append(path)
        if cfg.get("test_compute_metric", False):
            raise NotImplementedError("test_compute_metric (run_valid: Chamfer / EMD metrics) is out of scope of the hot path")
        return paths
