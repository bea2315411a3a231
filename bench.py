#!/usr/bin/env python
"""bench.py -- rollout throughput of the DMCF per-step hot path on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line on rank 0.
A "step" is one pass of the hot path (Simulator.run_inference: neighbour searches + 18 CConv/ASCC layers
of the Liquid3d SymNet with the reference's trained weights + integration) over one scene held in HBM.

Workload (BASELINE.json: metric quoted on "1M particles"; config 5 "synthetic 3-D box"): per GPU a cube of
``side``^3 = 1,000,000 fluid particles, spacing h = 0.05, jitter U(-0.1h, 0.1h) seed 0, velocities N(0, 0.1^2)
seed 1, closed 2-layer boundary shell (124,864 particles).  N > 1: one process per GPU, each with its own
box of the same size (weak scaling), no data-path collective in this round.

Extra objects on the JSON line:
  roofline      the CConv kernel (dmcf::cconv_kernel): algorithmic bytes (SURVEY.md section 8d formula) of all its
                launches in the timed steps / their summed duration measured with HIP events on the launch stream
  cpu_baseline  the CPU oracle (numpy + C restatement, OpenMP on all host cores) timed on a bounded sample of the
                same workload (a smaller box of the same density and network), rank 0, N = 1 only
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def cconv_algorithmic_bytes(m):
    """P*(4 idx + 4 dist + 12 xyz + 4*Cin feats) + n_out*(8 + 12 + 4*Cout) + 4*K*Cin*Cout."""
    return (m["pairs"] * (4 + 4 + 12 + 4 * m["cin"]) + m["n_out"] * (8 + 12 + 4 * m["cout"])
            + 4 * m["K"] * m["cin"] * m["cout"])


def cpu_baseline(side, weights, cfg):
    """Time ONE step of the CPU oracle (the restated reference path) on a side^3 box of the same density."""
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
                sample=f"1 step of the CPU oracle (oracle/model_ref.py, OpenMP) on a {side}^3 = {n}-particle box of the "
                       f"same density and network; {dt:.1f} s, {ref.pairs} neighbour pairs")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)  # the caching allocator reaches its steady state after ~3 steps
    ap.add_argument("--side", type=int, default=100, help="fluid cube edge in particles (100 -> 1M particles)")
    ap.add_argument("--cpu-side", type=int, default=40, help="edge of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--layers-json", default=None, help="write the per-launch table here")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # DMCF_BENCH_SHARDED=1 under torch.distributed.run with ONE process: the sharded driver + RCCL collectives at world
    # size 1 (the only way to exercise that path on a 1-GPU box; its ghost sets are empty)
    sharded = world > 1 or (os.environ.get("DMCF_BENCH_SHARDED") == "1" and "RANK" in os.environ)
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif args.gpus != 1:
        raise SystemExit("for --gpus N > 1 launch with torch.distributed.run (one process per GPU)")
    dev = torch.device("cuda", local_rank)

    from dmcf_amd import models, ops
    from dmcf_amd.pipelines import Simulator
    from dmcf_amd.utils import tf_checkpoint as tc
    from tools import configs, scenes

    cfg = configs.LIQUID3D
    weights = dict(np.load(os.path.join(ROOT, "tests", "golden", "liquid3d_weights.npz")))
    model = getattr(models, cfg["name"])(**cfg)
    tc.load_into_model(model, weights, device=dev)
    if not sharded:
        sim = Simulator(model, device=f"cuda:{local_rank}")
        scene = scenes.box_scene(args.side)
        n_fluid = scene["pos"].shape[0]
        state = scenes.model_inputs(scene, device=dev)
        step = lambda st: sim.step([st])[0]  # noqa: E731
    else:
        # weak scaling: a (side*N) x side x side box, rank r owns the r-th cube (slab along x); every layer
        # refreshes its ghost features with one all-to-all-v over RCCL (dmcf_amd/parallel.py)
        from dmcf_amd import parallel
        comm = parallel.TorchDistComm()
        decomp = parallel.SlabDecomposition.uniform(0, 0.0, world * args.side * 0.05, world)
        ssim = parallel.ShardedSimulator(model, comm, decomp)
        scene = scenes.box_slab_scene(args.side, world, rank)
        n_fluid = scene["pos"].shape[0]
        state = parallel.shard_scene(scene, decomp, rank, dev)
        state["gid"] = state["gid"] + rank * n_fluid
        step = ssim.step

    def barrier():
        torch.cuda.synchronize(dev)
        if sharded:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        state = step(state)
    if os.environ.get("DMCF_BENCH_DEBUG"):
        torch.cuda.reset_peak_memory_stats(dev)
    if rank == 0 and os.environ.get("DMCF_BENCH_NOTIMER") != "1":
        ops.timer = ops.LaunchTimer()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        state = step(state)
        if os.environ.get("DMCF_BENCH_DEBUG"):
            torch.cuda.synchronize(dev)
            ms = torch.cuda.memory_stats(dev)
            print(f"[debug] step done at {1e3 * (time.perf_counter() - t0):.1f} ms; reserved {ms['reserved_bytes.all.current'] / 2**30:.1f} GiB "
                  f"(peak {ms['reserved_bytes.all.peak'] / 2**30:.1f}), allocated peak {ms['allocated_bytes.all.peak'] / 2**30:.1f} GiB, "
                  f"device mallocs {ms['num_device_alloc']}, frees {ms['num_device_free']}, retries {ms['num_alloc_retries']}", file=sys.stderr, flush=True)
    barrier()
    elapsed = time.perf_counter() - t0
    timer, ops.timer = (ops.timer if ops.timer is not None else ops.LaunchTimer()), None
    assert torch.isfinite(state["pos"] if isinstance(state, dict) else state[0]).all()
    if sharded:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        recs = timer.results()
        conv = [(m, ms) for k, m, ms in recs if k == "cconv"]
        conv_ms = sum(ms for _, ms in conv)
        conv_bytes = sum(cconv_algorithmic_bytes(m) for m, _ in conv)
        achieved = conv_bytes / (conv_ms * 1e-3) / 1e9 if conv_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_cconv_hbm_traffic.json")
        if os.path.exists(pmc):
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        other = {}
        for k, m, ms in recs:
            other[k] = other.get(k, 0.0) + ms
        line = {
            "metric": "rollout_particle_steps_per_sec", "value": world * n_fluid * args.steps / elapsed,
            "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synthetic 3-D box, {n_fluid} fluid + {scene['box'].shape[0]} boundary particles per GPU, "
                                   f"Liquid3d SymNet (18 CConv/ASCC layers, reference checkpoint weights), one rollout step",
                       "parallelism": (f"{world} slabs along x of one {world * args.side}x{args.side}x{args.side} box, 1 process per GPU, "
                                       "per-layer ghost all-to-all-v over RCCL") if sharded else "single GPU",
                       "particles_per_gpu": n_fluid},
            "roofline": {"bound": "hbm", "kernel": "dmcf::cconv_* (all CConv/ASCC launches of the timed steps: cconv_kernel, cconv_mfma_kernel, cconv_blk_kernel, cconv_cls_kernel, cconv_z3_kernel, cconv_direct_kernel, lat_conv_kernel)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "launches": len(conv), "avg_launch_ms": conv_ms / max(len(conv), 1),
                         "algorithmic_bytes_per_launch": conv_bytes / max(len(conv), 1)},
            "kernel_ms_per_step": {k: v / args.steps for k, v in other.items()},
        }
        if world == 1 and args.cpu_side > 0:
            line["cpu_baseline"] = cpu_baseline(args.cpu_side, weights, cfg)
        if args.layers_json:
            json.dump([dict(kind=k, ms=ms, **m) for k, m, ms in recs], open(args.layers_json, "w"))
        print(json.dumps(line), flush=True)
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
