#!/usr/bin/env python
"""bench.py -- rollout throughput of the DMCF per-step hot path on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line on rank 0.
A "step" is one pass of the hot path (Simulator.run_inference: neighbour searches + 18 CConv/ASCC layers
of the Liquid3d SymNet with the reference's trained weights + integration) over one scene held in HBM.

Workload (BASELINE.json: metric quoted on "1M particles"; config 5 "synthetic 3-D box"): per GPU a cube of
``side``^3 = 1,000,000 fluid particles, spacing h = 0.05, jitter U(-0.1h, 0.1h) seed 0, velocities N(0, 0.1^2)
seed 1, closed 2-layer boundary shell (124,864 particles at N = 1).

``--gpus N`` with N > 1 and no RANK in the environment: bench.py launches its own N ranks (it re-executes itself
under ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1``); started BY such a
launcher (RANK / WORLD_SIZE set) it is one rank: one process per GPU, RCCL (backend "nccl") world size asserted = N.
The ranks shard ONE box of N * side^3 particles into axis-aligned blocks (dmcf_amd/parallel.py: 2 = 2x1x1, 4 = 2x2x1,
8 = 2x2x2; other N: slabs) -- weak scaling, side^3 particles per GPU -- and every CConv layer refreshes its ghost
features with one all-to-all-v over RCCL.

Extra objects on the JSON line:
  roofline          the DOMINANT CConv kernel of the timed steps (the template instantiation with the largest summed
                    duration): algorithmic bytes of its launches / their summed duration, HIP
                    events on the launch stream.  ``traffic`` cites the PMC measurement committed under profiles/ for
                    that kernel (``traffic_source``), it is not re-measured in this run.
  roofline_groups   the same fraction for (a) all neighbour-list kernels together, (b) the lattice-form launches --
                    charged the bytes THEY move (input volume + per-offset matrices + table + outputs), not the pair
                    bytes of a list they never read --, and per kernel instantiation
  cpu_baseline      the CPU oracle (numpy + C restatement, OpenMP on all host cores) timed on ONE step of the bench's own
                    scene ("same inputs"; ~1 minute at 1M particles; --cpu-side bounds it), rank 0, N = 1
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3  # dense f32-input MFMA (v_mfma_f32_16x16x4_f32 / 32x32x2_f32), same guide


def lattice_algorithmic_flops(m):
    """The lattice form (dmcf_amd/csrc/cconv_lat.hip) is a 3-D convolution: one [Cin x Cout] matrix per stencil offset and
    output point, 2 * n_out * offsets * Cin * Cout flops per launch (the offsets of all parts of a batched launch are summed
    in ``n_offsets``, each part covering its share of the outputs: n_out * n_offsets / parts = n_out_part * offsets summed)."""
    return 2.0 * m["n_out"] * m["n_offsets"] / max(m.get("parts", 1), 1) * m["cin"] * m["cout"]


def cconv_algorithmic_bytes(m):
    """SURVEY.md section 8d, no cache credit.  Neighbour-list launch: P * (4 index + 12 neighbour xyz + 4 * Cin features
    [+ 4 importance / d^2 only when the list carries that array]) + n_out * (8 row split + 12 xyz + 4 * Cout) +
    4 * K * Cin * Cout.  Lattice launch (no list): the input volume, the per-offset matrices, the cell table, the outputs.
    A launch that ADDS to its output (DMCF_FLAG_ACCUMULATE: the layer sums of models/hrnet.py) reads it first: + n_out * 4 * Cout,
    the bytes of the elementwise kernel it replaces."""
    rmw = m["n_out"] * 4 * m["cout"] if m.get("accumulate") else 0
    if m.get("lattice"):
        return (m["volume_bytes"] + m["table_bytes"] + 4 * m["n_offsets"] * m["cin"] * m["cout"]
                + m["n_out"] * 4 * m["cout"] + 4 * m["K"] * m["cin"] * m["cout"] + rmw)
    per_pair = 4 + 12 + 4 * m["cin"] + (4 if m.get("pair_values", True) else 0)
    return m["pairs"] * per_pair + m["n_out"] * (8 + 12 + 4 * m["cout"]) + 4 * m["K"] * m["cin"] * m["cout"] + rmw


def cconv_algorithmic_flops(m):
    """P (2 * 8 Cin [trilinear splat] + 60 [mapping + window]) + n_out * 2 K Cin Cout [contraction] -- what a
    neighbour-list launch has to compute whatever its form; against the f32 matrix peak it is the bound the byte model is not (the
    kernels' DRAM traffic is an eighth of their algorithmic bytes: they run out of L2)."""
    return m["pairs"] * (16.0 * m["cin"] + 60.0) + m["n_out"] * 2.0 * m["K"] * m["cin"] * m["cout"]


def frs_algorithmic_bytes(m):
    """12 (n_in + n_out) + 12 n_in + P * (4 [+ 4 distances]) + 8 n_out."""
    return (12 * (m["n_points"] + m["n_queries"]) + 12 * m["n_points"] + m.get("pairs", 0) * (8 if m.get("distances") else 4)
            + 8 * m["n_queries"])


def cpu_baseline(side, weights, cfg, same=False):
    """Time ONE step of the CPU oracle (the restated reference path) on a side^3 box from the bench's scene generator --
    ``same``: the very scene the GPU steps (its state before the first step)."""
    import oracle  # noqa: F401  (builds the C library if needed)
    from oracle.model_ref import ModelRef
    from tools import scenes
    scene = scenes.box_scene(side)
    ref = ModelRef(cfg, weights)
    data = scenes.model_inputs(scene)
    t0 = time.time()
    ref.step(data)
    dt = time.time() - t0
    n = scene["pos"].shape[0]
    return dict(value=n / dt, unit="particle-steps/s", cores=len(os.sched_getaffinity(0)), kind="port",
                sample=f"1 step of the CPU oracle (oracle/model_ref.py: numpy + OpenMP C restatement of the Open3D CPU "
                       f"algorithms; not TensorFlow/Open3D) on " + ("the bench's own scene, the state before its first step: the " if same else "a bounded sample: a ")
                       + f"{side}^3 = {n}-particle box (+ {scene['box'].shape[0]} boundary) from the bench's scene generator, same density "
                       f"and network; {dt:.1f} s, {ref.pairs} neighbour pairs")


def scene_snapshot(pos, vel, lo, hi):
    """What state the scene is in (outside the timed region): fluid particles outside the boundary shell's bounding box, the
    largest speed.  The 5 m column of config 5 is far outside what the network was trained on: from step ~10 on particles leak
    through the 2-layer shell, rows get longer and a step costs more -- the line says so."""
    import torch
    out = ((pos < lo) | (pos > hi)).any(dim=1).sum()
    return [int(out.item()), float(vel.norm(dim=1).max().item()) if pos.shape[0] else 0.0]


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(argv, gpus, port=None):
    """The command ``python bench.py --gpus N`` re-executes itself with (one rank per GPU)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), os.path.abspath(__file__)] + list(argv)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)  # the caching allocator reaches its steady state after ~3 steps
    ap.add_argument("--side", type=int, default=100, help="fluid cube edge in particles per GPU (100 -> 1M particles)")
    ap.add_argument("--cpu-side", type=int, default=None,
                    help="edge of the CPU-baseline box (default: --side, i.e. ONE step of the CPU oracle on the bench's own scene -- "
                         "about a minute of the host's cores at 1M particles; smaller = a bounded sample from the same generator; 0 = skip)")
    ap.add_argument("--config", default="box", choices=["box", "waterramps", "wbcsph", "liquid3d_dam"],
                    help="box (default) = BASELINE.json config 5, the headline; the others = configs 2 / 3 / 4 on one GPU "
                         "(tools/long_rollout.py's scenes): the same JSON line with dispatches and synchronising reads per step")
    ap.add_argument("--scene", default="column", choices=["column", "settled"],
                    help="--config box only.  column (default, the headline): config 5 as specified -- a 5 m column under g, far outside "
                         "the network's training range, which dissolves inside the window.  settled: the same box, particles and "
                         "network, but every timed step starts from the SAME state (the one the warm-up steps end in): the work "
                         "the kernels are timed on does not change from step to step.  (A physically settled column was tried -- "
                         "particles at rest, gravity scaled to 0 ... 1 %, velocities damped to zero after every step, thicker and "
                         "denser shells: the network alone pushes tens of thousands of particles of this jittered lattice through "
                         "a synthetic wall within 25 steps, DESIGN.md section 5)")
    ap.add_argument("--layers-json", default=None, help="write the per-launch table here")
    ap.add_argument("--reserve-gib", type=float, default=None,
                    help="Simulator(reserve_gib=...) in GiB (default: the product's opt-in 'auto' rule, 40 KiB per particle handed "
                         "to the caching allocator before the first step; 0 = none)")
    ap.add_argument("--decomp", default="blocks", choices=["blocks", "slabs"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = side^3 particles PER GPU (one box of N side^3, the default the driver's efficiency is "
                         "computed from); strong = ONE box of side^3 particles split over the N GPUs (BASELINE.json's '1M particles "
                         "@1/2/4/8 GPUs' read literally)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch / rendezvous / reduction logic only, on the gloo backend without a GPU (CPU test)")
    return ap.parse_args(argv)


def block_grid(world):
    """Ranks per axis (x, y, z) of the block decomposition: SURVEY.md section 8e (8 = 2x2x2, 4 = 2x2x1, 2 = 2x1x1)."""
    grid = [1, 1, 1]
    k, w = 0, world
    while w % 2 == 0 and w > 1:
        grid[k % 3] *= 2
        w //= 2
        k += 1
    grid[0] *= w  # an odd factor: slabs along x
    return grid


def summarise(recs, steps):
    """Per-kernel roofline table from the (kind, meta, ms) launch records of the timed steps."""
    groups = {}
    for kind, m, ms in recs:
        if kind != "cconv":
            continue
        g = groups.setdefault(m.get("kernel", "cconv"), dict(launches=0, ms=0.0, bytes=0, flops=0.0, lattice=bool(m.get("lattice"))))
        g["launches"] += 1
        g["ms"] += ms
        g["bytes"] += cconv_algorithmic_bytes(m)
        g["flops"] += lattice_algorithmic_flops(m) if m.get("lattice") else cconv_algorithmic_flops(m)

    def frac(gs):
        ms = sum(g["ms"] for g in gs)
        by = sum(g["bytes"] for g in gs)
        n = sum(g["launches"] for g in gs)
        gbs = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        d = dict(launches=n, ms_per_step=ms / steps, avg_launch_ms=ms / max(n, 1), algorithmic_bytes_per_launch=by / max(n, 1),
                 achieved=gbs, frac=gbs / HBM_PEAK_GBS)
        fl = sum(g["flops"] for g in gs)
        tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        if gs and all(g["lattice"] for g in gs):
            # the lattice form: a dense 3-D convolution on the f32 matrix cores -- its own roofline next to the byte one
            d.update(bound="mfma_f32", algorithmic_flops_per_launch=fl / max(n, 1), achieved_tflops=tf, peak_tflops=MFMA_F32_PEAK_TFLOPS,
                     frac_mfma_f32=tf / MFMA_F32_PEAK_TFLOPS)
        elif fl > 0:
            # neighbour-list kernels: the contract's byte fraction above (`frac`), the same launches against the f32 matrix peak
            # (`frac_flops`: SURVEY 8d's flops; the splat's share would run on the vector pipe at 1/2 of that peak at best, so
            # this is an upper bound on how close to compute-bound they are) ...
            d.update(algorithmic_flops_per_launch=fl / max(n, 1), achieved_tflops=tf, peak_tflops=MFMA_F32_PEAK_TFLOPS,
                     frac_flops=tf / MFMA_F32_PEAK_TFLOPS)
        return d
    nl = [g for g in groups.values() if not g["lattice"]]
    lat = [g for g in groups.values() if g["lattice"]]
    table = dict(neighbour_list=frac(nl), lattice=frac(lat), by_kernel={k: frac([g]) for k, g in sorted(groups.items())})
    # ... and the DRAM traffic the counters measured for the kernel (profiles/*hbm_traffic.json, per launch) over ITS launch time
    # here against the HBM peak (`frac_dram`): what the memory system actually moves
    for k, d in table["by_kernel"].items():
        tb, src = cited_traffic(k)
        if tb and d["avg_launch_ms"] > 0:
            d.update(traffic=tb, traffic_source=src, frac_dram=tb / (d["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS)
    # (the roofline object is the dominant NEIGHBOUR-LIST kernel: the lattice launches are matrix-pipe bound by design and are
    # charged their own, much smaller byte count in roofline_groups.lattice)
    nl_names = [k for k in groups if not groups[k]["lattice"]] or list(groups)
    dominant = max(nl_names, key=lambda k: groups[k]["ms"]) if nl_names else None
    return table, dominant


def cited_traffic(kernel):
    """HBM bytes per launch of ``kernel`` from the newest PMC summary committed under profiles/ (FETCH_SIZE / WRITE_SIZE
    passes of the bench command, corrected as MI355X_MICROARCH.md prescribes); (None, None) if there is none."""
    best = (None, None)
    pdir = os.path.join(ROOT, "profiles")
    for f in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if not f.endswith("hbm_traffic.json"):
            continue
        try:
            d = json.load(open(os.path.join(pdir, f)))
        except ValueError:
            continue
        per = d.get("by_kernel", {})
        for name, v in per.items():
            if kernel and (name == kernel or name.endswith("::" + kernel) or kernel in name):
                best = (v.get("hbm_bytes_per_launch"), f"profiles/{f} [{name}]")
    return best


SMALL_CONFIGS = {
    "waterramps": "BASELINE.json config 2 (WaterRamps 2-D, ~2k particles, 600-step rollout): the WaterRamps SymNet (configs/WaterRamps.yml) "
                  "on a 45 x 45 = 2025-particle 2-D box + shell, seeded stand-in weights (the reference's blob is missing)",
    "wbcsph": "BASELINE.json config 3 (WBC-SPH 2-D, 3200-step rollout): the WBC-SPH SymNet (configs/WBC-SPH.yml, 4 scales, grav_eqvar) on a "
              "60 x 60 = 3600-particle 2-D box + shell, seeded stand-in weights (the reference's blob is missing)",
    "liquid3d_dam": "BASELINE.json config 4 (Liquid3d 3-D, ~100k particles): a 50 x 40 x 50 = 100,000-particle dam break in an open tank, "
                    "Liquid3d SymNet with the reference's checkpoint weights",
}


def small_config_main(args):
    """``--config waterramps | wbcsph | liquid3d_dam``: BASELINE.json configs 2 / 3 / 4 on one GPU, the same JSON line.  These
    steps are paced by dispatch count and host reads, not by any kernel, so the line also carries the GPU kernel dispatches and
    the synchronising device -> host reads of ONE step (counted on an extra step after the timed window).  The timed window
    runs as Simulator.run_rollout runs its loop (inside steady_steps) WITHOUT the per-launch events; the per-kernel table comes
    from a second, untimed pass of the same length."""
    import warnings

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the DMCF hot path has no CPU fallback)")
    from dmcf_amd import models, ops
    from dmcf_amd.pipelines import Simulator
    from dmcf_amd.pipelines.simulator import steady_steps
    from dmcf_amd.utils import tf_checkpoint as tc
    from tools import long_rollout, scenes
    dev = torch.device("cuda", 0)
    cfg, w, scene, grav = long_rollout.setup(args.config)
    model = getattr(models, cfg["name"])(**cfg)
    tc.load_into_model(model, w, device=dev)
    sim = Simulator(model, device="cuda:0", reserve_gib="auto" if args.reserve_gib is None else args.reserve_gib)
    state = scenes.model_inputs(scene, device=dev, grav=grav)
    n = int(state[0].shape[0])
    lo = torch.tensor(scene["box"].min(axis=0), device=dev)
    hi = torch.tensor(scene["box"].max(axis=0), device=dev)
    if args.config == "liquid3d_dam":
        hi[1] = 3.0e38  # the tank is open at the top
    if scene["box"][:, 2].max() == scene["box"][:, 2].min():
        lo[2], hi[2] = -3.0e38, 3.0e38  # 2-D scenes
    for _ in range(args.warmup):
        state = sim.step([state])[0]
    snap0 = scene_snapshot(state[0], state[1], lo, hi)
    rep0 = sim.repeated_steps
    allocs0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    with steady_steps() as steady:
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            state = sim.step([state])[0]
            steady.tick()
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
        snap1 = scene_snapshot(state[0], state[1], lo, hi)
        repeated = sim.repeated_steps - rep0
        allocs = torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - allocs0
        assert torch.isfinite(state[0]).all()
        # ---- untimed: one step's synchronising reads and kernel dispatches
        torch.cuda.set_sync_debug_mode("warn")
        with warnings.catch_warnings(record=True) as ws:
            warnings.simplefilter("always")
            state = sim.step([state])[0]
        torch.cuda.set_sync_debug_mode("default")
        sync_sites = {}
        for wv in ws:
            k = f"{os.path.basename(wv.filename)}:{wv.lineno}"
            sync_sites[k] = sync_sites.get(k, 0) + 1
        dispatches = None
        try:
            from torch.profiler import ProfilerActivity, profile
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                state = sim.step([state])[0]
                torch.cuda.synchronize(dev)
            kinds = {}
            for e in prof.events():
                if getattr(e, "device_type", None) is not None and "cuda" in str(e.device_type).lower():
                    kinds[e.name] = kinds.get(e.name, 0) + 1
            memops = sum(v for k, v in kinds.items() if k.lower().startswith(("memcpy", "memset")))
            dispatches = dict(kernels=sum(kinds.values()) - memops, memcpy_memset=memops)
        except Exception as e:  # (the count is a diagnostic: never fail the line for it)
            dispatches = dict(error=f"{type(e).__name__}: {e}")
        # ---- untimed: the same number of steps with HIP events around every library launch -> per-kernel table, per-step times
        ops.timer = ops.LaunchTimer()
        marks, step_ms = [], []
        for _ in range(args.steps):
            marks.append(len(ops.timer.records))
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            state = sim.step([state])[0]
            torch.cuda.synchronize(dev)
            step_ms.append(1e3 * (time.perf_counter() - t1))
            steady.tick()
        timer, ops.timer = ops.timer, None
    recs = timer.results()
    table, dominant = summarise(recs, args.steps)
    dom = table["by_kernel"].get(dominant, dict(achieved=0.0, frac=0.0, launches=0, avg_launch_ms=0.0, algorithmic_bytes_per_launch=0.0))
    other = {}
    for k, m, ms in recs:
        other[k] = other.get(k, 0.0) + ms
    line = {
        "metric": "rollout_particle_steps_per_sec", "value": n * args.steps / elapsed, "unit": "particle-steps/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": SMALL_CONFIGS[args.config] + f"; {n} fluid + {int(state[4].shape[0])} boundary particles, one rollout step",
                   "parallelism": "single GPU", "particles_per_gpu": n},
        "roofline": {"bound": "hbm", "kernel": f"dmcf::{dominant}", "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": dom["frac"], "traffic": None, "launches": dom["launches"], "avg_launch_ms": dom["avg_launch_ms"],
                     "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"], "frac_flops": dom.get("frac_flops"),
                     "note": "from the instrumented pass; a step of this size is bound by dispatches and host reads, not by this kernel"},
        "roofline_groups": table,
        "kernel_ms_per_step": {k: v / args.steps for k, v in other.items()},
        "library_launch_ms_per_step": sum(other.values()) / args.steps,
        "dispatches_per_step": dispatches,
        "synchronising_reads_per_step": {"count": len(ws), "sites": sync_sites},
        "step_ms_instrumented": {"median": float(np.median(step_ms)), "first": float(step_ms[0]), "last": float(step_ms[-1]),
                                 "note": "second pass, one device synchronise per step and events around every library launch"},
        "scene_state": {"simulated_steps_before_window": args.warmup, "simulated_steps_at_end": args.warmup + args.steps,
                        "fluid_outside_shell_at_window_start": snap0[0], "fluid_outside_shell_at_end": snap1[0],
                        "max_speed_at_window_start": snap0[1], "max_speed_at_end": snap1[1],
                        "repeated_steps_in_window": repeated, "device_allocations_in_window": int(allocs),
                        "reserved_gib": torch.cuda.memory_stats(dev)["reserved_bytes.all.current"] / 2 ** 30},
    }
    if args.cpu_side is None or args.cpu_side > 0:
        # the CPU oracle on the same scene from its initial state: whole steps until ~10 s are spent (at most 40)
        import oracle  # noqa: F401
        from oracle.model_ref import ModelRef
        ref = ModelRef(cfg, w)
        data = scenes.model_inputs(scene, grav=grav)
        t1, k = time.time(), 0
        while k < 40 and (k == 0 or time.time() - t1 < 10.0):
            p, v = ref.step(data)
            data = [p, v] + data[2:]
            k += 1
        dt = time.time() - t1
        line["cpu_baseline"] = dict(value=n * k / dt, unit="particle-steps/s", cores=len(os.sched_getaffinity(0)), kind="port",
                                    sample=f"{k} consecutive step(s) of the CPU oracle (oracle/model_ref.py: numpy + OpenMP C restatement "
                                           f"of the Open3D CPU algorithms; not TensorFlow/Open3D) on the same scene from its initial state, {dt:.1f} s")
    if args.layers_json:
        json.dump([dict(kind=k, ms=ms, **m) for k, m, ms in recs], open(args.layers_json, "w"))
    print(json.dumps(line), flush=True)


def main():
    args = parse_args()
    in_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.config != "box":
        if args.gpus != 1:
            raise SystemExit("--config waterramps / wbcsph / liquid3d_dam are single-GPU lines")
        return small_config_main(args)
    if args.scene == "settled" and args.gpus != 1:
        raise SystemExit("--scene settled is a single-GPU line")
    if args.gpus > 1 and not in_launcher:
        # the driver's contract: plain `python bench.py --gpus N` -- start the N ranks ourselves
        raise SystemExit(subprocess.call(launcher_command(sys.argv[1:], args.gpus)))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if in_launcher and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    # DMCF_BENCH_SHARDED=1 under a launcher with ONE process: the sharded driver + RCCL collectives at world size 1 (the
    # only way to exercise that path on a 1-GPU box; its ghost sets are empty)
    sharded = world > 1 or (os.environ.get("DMCF_BENCH_SHARDED") == "1" and in_launcher)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

    if args.dry_run:
        # everything around the step: rendezvous, world-size check, barrier, max-over-ranks timing, one line from rank 0
        if sharded:
            dist.init_process_group("gloo")
            assert dist.get_world_size() == args.gpus and dist.get_rank() == rank
        t0 = time.perf_counter()
        time.sleep(0.01 * (rank + 1))
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        if sharded:
            dist.barrier()
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"metric": "rollout_particle_steps_per_sec", "dry_run": True, "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "block_grid": block_grid(world), "max_rank_seconds": float(el.item())}), flush=True)
        if sharded:
            dist.destroy_process_group()
        return

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the DMCF hot path has no CPU fallback); --dry-run tests the launch logic")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: local rank {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    if sharded:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == world == args.gpus, "RCCL world size != --gpus"
    dev = torch.device("cuda", local_rank)

    from dmcf_amd import models, ops
    from dmcf_amd.pipelines import Simulator
    from dmcf_amd.utils import tf_checkpoint as tc
    from tools import configs, scenes

    cfg = configs.LIQUID3D
    weights = dict(np.load(os.path.join(ROOT, "tests", "golden", "liquid3d_weights.npz")))
    model = getattr(models, cfg["name"])(**cfg)
    tc.load_into_model(model, weights, device=dev)
    extra = {}
    # (opt-in in the product since round 5; the bench asks for it -- a rollout of this size otherwise meets multi-GB hipMallocs
    # in its first steps -- and reports what it got: scene_state.reserved_gib)
    sim_kw = dict(reserve_gib="auto" if args.reserve_gib is None else args.reserve_gib)
    if not sharded:
        sim = Simulator(model, device=f"cuda:{local_rank}", **sim_kw)
        settled = args.scene == "settled"
        scene = scenes.box_scene(args.side)
        n_fluid = scene["pos"].shape[0]
        n_total = n_fluid
        state = scenes.model_inputs(scene, device=dev)
        step = lambda st: sim.step([st])[0]  # noqa: E731
        par = "single GPU"
    else:
        # weak scaling: ONE box of grid * side particles, rank r owns one block of side^3; every layer refreshes its ghost
        # features with one all-to-all-v over RCCL (dmcf_amd/parallel.py)
        from dmcf_amd import parallel
        comm = parallel.TorchDistComm()
        grid = block_grid(world) if args.decomp == "blocks" else [world, 1, 1]
        h = 0.05
        if args.scaling == "strong":
            if any(args.side % g for g in grid):
                raise SystemExit(f"--scaling strong: --side {args.side} is not divisible by the block grid {grid}")
            sides = [args.side // g for g in grid]
        else:
            sides = [args.side] * 3
        decomp = parallel.BlockDecomposition.uniform([0.0, 0.0, 0.0], [g * sd * h for g, sd in zip(grid, sides)], grid)
        ssim = parallel.ShardedSimulator(model, comm, decomp, **sim_kw)
        parallel.stats().profile = True  # HIP events around every wait for a ghost exchange (outside of nothing: they are the step)
        scene = scenes.box_block_scene(sides, grid, rank)
        n_fluid = scene["pos"].shape[0]
        state = parallel.shard_scene(scene, decomp, rank, dev, presharded=True)
        state["gid"] = state["gid"] + rank * n_fluid
        n_total = world * n_fluid
        step = ssim.step
        par = (f"{grid[0]}x{grid[1]}x{grid[2]} blocks of one {grid[0] * sides[0]}x{grid[1] * sides[1]}x{grid[2] * sides[2]} box "
               f"({args.scaling} scaling), 1 process per GPU, per-layer ghost all-to-all-v over RCCL")

    def barrier():
        torch.cuda.synchronize(dev)
        if sharded:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        state = step(state)
    _pv = (lambda st: (st["pos"], st["vel"]) if isinstance(st, dict) else (st[0], st[1]))
    shell_lo = torch.tensor(scene["box"].min(axis=0), device=dev) if scene["box"].shape[0] else None
    shell_hi = torch.tensor(scene["box"].max(axis=0), device=dev) if scene["box"].shape[0] else None
    if sharded:  # the shell of the WHOLE box: every rank holds a piece of it
        big = 3.0e38
        lohi = torch.cat([-(shell_lo if shell_lo is not None else torch.full((3,), big, device=dev)),
                          shell_hi if shell_hi is not None else torch.full((3,), -big, device=dev)])
        dist.all_reduce(lohi, op=dist.ReduceOp.MAX)
        shell_lo, shell_hi = -lohi[:3], lohi[3:]
    snap0 = scene_snapshot(*_pv(state), shell_lo, shell_hi)
    allocs0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    repeated0 = 0 if sharded else sim.repeated_steps
    if os.environ.get("DMCF_BENCH_DEBUG"):
        torch.cuda.reset_peak_memory_stats(dev)
    if rank == 0 and os.environ.get("DMCF_BENCH_NOTIMER") != "1":
        ops.timer = ops.LaunchTimer()
    barrier()
    t0 = time.perf_counter()
    settled_scene = (not sharded) and args.scene == "settled"
    settled_out = None
    step_marks = []
    step_events = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]  # (device time stamps; nothing waits for them)
    step_events[0].record()
    for i in range(args.steps):
        if ops.timer is not None:
            step_marks.append(len(ops.timer.records))
        if settled_scene:
            settled_out = step(state)  # (the same input every time; the output is kept for the closing checks only)
        else:
            state = step(state)
        step_events[i + 1].record()
        if os.environ.get("DMCF_BENCH_DEBUG"):
            torch.cuda.synchronize(dev)
            ms = torch.cuda.memory_stats(dev)
            print(f"[debug] step done at {1e3 * (time.perf_counter() - t0):.1f} ms; reserved {ms['reserved_bytes.all.current'] / 2**30:.1f} GiB "
                  f"(peak {ms['reserved_bytes.all.peak'] / 2**30:.1f}), allocated peak {ms['allocated_bytes.all.peak'] / 2**30:.1f} GiB, "
                  f"device mallocs {ms['num_device_alloc']}, frees {ms['num_device_free']}, retries {ms['num_alloc_retries']}", file=sys.stderr, flush=True)
    rank_elapsed = time.perf_counter() - t0  # (this rank's own clock, before the closing barrier)
    barrier()
    elapsed = time.perf_counter() - t0
    timer, ops.timer = (ops.timer if ops.timer is not None else ops.LaunchTimer()), None
    if settled_scene:
        state = settled_out
    assert torch.isfinite(state["pos"] if isinstance(state, dict) else state[0]).all()
    from dmcf_amd.utils.convolutions import neighbor_hints
    snap1 = scene_snapshot(*_pv(state), shell_lo, shell_hi)
    allocs = torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - allocs0
    if sharded:
        t2 = torch.tensor([snap0[0], snap1[0], allocs], dtype=torch.int64, device=dev)
        dist.all_reduce(t2)
        v2 = torch.tensor([snap0[1], snap1[1]], dtype=torch.float64, device=dev)
        dist.all_reduce(v2, op=dist.ReduceOp.MAX)
        snap0, snap1, allocs = [int(t2[0]), float(v2[0])], [int(t2[1]), float(v2[1])], int(t2[2])
    step_ms = [step_events[i].elapsed_time(step_events[i + 1]) for i in range(args.steps)]
    q = max(args.steps // 4, 1)
    extra["step_ms"] = {"first_quarter": float(np.mean(step_ms[:q])), "last_quarter": float(np.mean(step_ms[-q:])),
                        "drift": float(np.mean(step_ms[-q:]) / np.mean(step_ms[:q]) - 1.0), "median": float(np.median(step_ms)),
                        "note": "device time between consecutive steps' ends on the launch stream (rank 0's)"}
    extra["scene_state"] = {
        "scene": args.scene if not sharded else "column",
        "simulated_steps_before_window": args.warmup, "simulated_steps_at_end": args.warmup + args.steps,
        "fluid_outside_shell_at_window_start": snap0[0], "fluid_outside_shell_at_end": snap1[0],
        "max_speed_at_window_start": snap0[1], "max_speed_at_end": snap1[1],
        "longest_rows_last_step": sorted({int(h) for h in neighbor_hints() if h is not None}),
        "repeated_steps_in_window": None if sharded else sim.repeated_steps - repeated0,
        "device_allocations_in_window": int(allocs),
        "reserved_gib": torch.cuda.memory_stats(dev)["reserved_bytes.all.current"] / 2 ** 30,
        "note": ("every timed step starts from the same state (the rollout's state after the warm-up steps): identical work per step, "
                 "for judging kernel work; the headline is --scene column") if (not sharded and args.scene == "settled") else
                "config 5's 5 m column is far outside the network's training range: particles leak through the 2-layer shell "
                "as the rollout goes on, rows lengthen and steps get slower (DESIGN.md section 4.1)"}
    if sharded:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # particles actually stepped: the sum over the ranks of what they own now (migration moves particles, none are lost)
        cnt = torch.tensor([state["pos"].shape[0]], dtype=torch.int64, device=dev)
        dist.all_reduce(cnt)
        assert int(cnt.item()) == n_total, f"particles lost in migration: {int(cnt.item())} != {n_total}"
        extra["ghost_rows_per_step_rank0"] = int(ssim.exchanged_rows / max(args.steps + args.warmup, 1))
        # what the step costs besides its kernels, per rank (all steps incl. warm-up for the row counts; the timed steps for the
        # waits): ghost rows received, rows migrated, device -> host reads, ms the compute stream stood still for an exchange
        st = parallel.stats()
        torch.cuda.synchronize(dev)
        mine = dict(rank=rank, fluid_particles=int(state["pos"].shape[0]), boundary_particles=int(state["box"].shape[0]),
                    ghost_rows_per_step=int(ssim.exchanged_rows / max(args.steps + args.warmup, 1)),
                    migrated_rows_per_step=ssim.migrated_rows_total / max(args.steps + args.warmup, 1),
                    host_syncs_per_step=st.host_syncs / max(args.steps + args.warmup, 1),
                    exposed_exchange_wait_ms_per_step=st.exposed_wait_ms() / max(args.steps, 1),
                    rank_seconds=rank_elapsed)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        extra["per_rank"] = per_rank

    if rank == 0:
        recs = timer.results()
        table, dominant = summarise(recs, args.steps)
        # the lattice launches over the window: the dense volumes cover the bounding box of the lattices, which grows with every
        # particle that leaves the shell -- first and last timed step side by side
        def lattice_step(i):
            lo, hi = step_marks[i], (step_marks[i + 1] if i + 1 < len(step_marks) else len(recs))
            ls = [(m, ms) for k, m, ms in recs[lo:hi] if k == "cconv" and m.get("lattice")]
            return dict(launches=len(ls), ms=sum(ms for _, ms in ls), volume_mb=sum(m["volume_bytes"] for m, _ in ls) / 1e6,
                        table_mb=sum(m["table_bytes"] for m, _ in ls) / 1e6, outputs=sum(m["n_out"] for m, _ in ls),
                        tflops=(sum(lattice_algorithmic_flops(m) for m, _ in ls) / (sum(ms for _, ms in ls) * 1e-3) / 1e12
                                if ls else 0.0))
        if step_marks:
            table["lattice"]["first_timed_step"] = lattice_step(0)
            table["lattice"]["last_timed_step"] = lattice_step(len(step_marks) - 1)
        dom = table["by_kernel"].get(dominant, dict(achieved=0.0, frac=0.0, launches=0, avg_launch_ms=0.0, algorithmic_bytes_per_launch=0.0))
        traffic, traffic_source = cited_traffic(dominant)
        other = {}
        for k, m, ms in recs:
            other[k] = other.get(k, 0.0) + ms
        frs = [(m, ms) for k, m, ms in recs if k in ("frs_search_padded", "frs_write", "frs_count")]
        frs_ms = sum(ms for _, ms in frs)
        frs_bytes = sum(frs_algorithmic_bytes(m) for m, _ in frs if "pairs" in m)
        line = {
            "metric": "rollout_particle_steps_per_sec", "value": n_total * args.steps / elapsed,
            "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling if sharded else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synthetic 3-D box (BASELINE.json config 5), {n_fluid} fluid particles per GPU + closed 2-layer "
                                   f"boundary shell ({scene['box'].shape[0]} boundary particles on rank 0), Liquid3d SymNet (18 CConv/ASCC "
                                   "layers, reference checkpoint weights), one rollout step"
                                   + ("; SETTLED variant: every timed step starts from the state after the warm-up" if (not sharded and args.scene == "settled") else ""),
                       "parallelism": par, "particles_per_gpu": n_fluid},
            "roofline": {"bound": "hbm", "kernel": f"dmcf::{dominant}", "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": dom["frac"], "traffic": traffic, "traffic_source": traffic_source,
                         "launches": dom["launches"], "avg_launch_ms": dom["avg_launch_ms"],
                         "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
                         # `frac` is the contract's figure (algorithmic bytes, no cache credit) and is NOT a bound for this
                         # kernel: its DRAM traffic is several times smaller (it runs out of L2).  The two fractions that bound:
                         "frac_flops": dom.get("frac_flops"), "algorithmic_flops_per_launch": dom.get("algorithmic_flops_per_launch"),
                         "achieved_tflops": dom.get("achieved_tflops"), "peak_tflops": MFMA_F32_PEAK_TFLOPS,
                         "frac_dram": (traffic / (dom["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                                       if traffic and dom["avg_launch_ms"] > 0 else None)},
            "roofline_groups": table,
            "search": {"ms_per_step": frs_ms / args.steps, "achieved": frs_bytes / (frs_ms * 1e-3) / 1e9 if frs_ms > 0 else 0.0,
                       "unit": "GB/s", "launches": len(frs)},
            "kernel_ms_per_step": {k: v / args.steps for k, v in other.items()},
        }
        line.update(extra)
        cpu_side = args.side if args.cpu_side is None else args.cpu_side
        if world == 1 and cpu_side > 0:
            line["cpu_baseline"] = cpu_baseline(cpu_side, weights, cfg, same=cpu_side == args.side)
        if args.layers_json:
            json.dump([dict(kind=k, ms=ms, **m) for k, m, ms in recs], open(args.layers_json, "w"))
        print(json.dumps(line), flush=True)
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
