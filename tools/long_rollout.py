"""Full-length rollouts of the configurations BASELINE.json names (README.md:79: 200 / 600 / 3200 steps), on the GPU:
every step is checked for finiteness and momentum conservation (the ASCC head: sum over fluid + boundary of the network
output = 0), the first steps against the CPU oracle fed with the HIP path's own states, and the rollout's memory behaviour
is logged: steps repeated after a NeighborCapacityExceeded, fresh device allocations (hipMalloc) per step, step times.

    python tools/long_rollout.py liquid3d_dam 200     # config 4: 100,000-particle dam break, Liquid3d weights
    python tools/long_rollout.py waterramps 600       # config 2 architecture, ~2k particles, seeded stand-in weights
    python tools/long_rollout.py wbcsph 3200          # config 3 architecture
Prints one JSON summary line (and writes per-step records with --out)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def setup(name):
    from tools import configs, scenes
    if name == "liquid3d_dam":
        cfg = configs.LIQUID3D
        w = dict(np.load(os.path.join(ROOT, "tests", "golden", "liquid3d_weights.npz")))
        return cfg, w, scenes.dam_break_scene(), None
    if name == "waterramps":
        cfg = configs.WATERRAMPS
        return cfg, scenes.random_weights(cfg, seed=0), scenes.box_scene(45, h=0.005, dim=2, origin=(-0.11, -0.11, 0.0), vel_std=0.0), None
    if name == "wbcsph":
        cfg = configs.WBC_SPH
        return cfg, scenes.random_weights(cfg, seed=1), scenes.box_scene(60, h=0.0025, dim=2, vel_std=0.0), np.float32([0.0, -9.81, 0.0])
    raise SystemExit(f"unknown rollout {name!r}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("steps", type=int)
    ap.add_argument("--oracle-steps", type=int, default=5)
    ap.add_argument("--out", default=None)
    ap.add_argument("--gc", action="store_true", help="diagnostic: leave Python's cyclic garbage collector on (the loop otherwise runs "
                                                      "as Simulator.run_rollout does: inside pipelines.simulator.steady_steps)")
    args = ap.parse_args()
    from dmcf_amd import models
    from dmcf_amd.pipelines import Simulator
    from dmcf_amd.utils import tf_checkpoint as tc
    from oracle.model_ref import ModelRef
    from tools import scenes
    dev = torch.device("cuda:0")
    cfg, w, scene, grav = setup(args.name)
    model = getattr(models, cfg["name"])(**cfg)
    tc.load_into_model(model, w, device=dev)
    sim = Simulator(model, device="cuda", reserve_gib="auto")
    ref = ModelRef(cfg, w)
    state = scenes.model_inputs(scene, device=dev, grav=grav)
    n = state[0].shape[0]
    recs, worst_parity, worst_mom = [], 0.0, 0.0
    allocs0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    t_all = time.time()
    for t in range(args.steps):
        if not args.gc and t == 4:
            from dmcf_amd.pipelines.simulator import steady_steps
            steady = steady_steps()
            steady.__enter__()
        if not args.gc and t > 4:
            steady.tick()
        before = [None if x is None else x.cpu().numpy() for x in state] if t < args.oracle_steps else None
        a0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
        rep0 = sim.repeated_steps
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        state = sim.step([state])[0]
        torch.cuda.synchronize(dev)
        ms = 1e3 * (time.perf_counter() - t0)
        pos = state[0]
        finite = bool(torch.isfinite(pos).all()) and bool(torch.isfinite(state[1]).all())
        out = torch.cat([model.pos_correction, model.obs], dim=0).double()
        mom = float((out.sum(0).abs() / out.abs().sum(0).clamp(min=1e-300)).max())
        worst_mom = max(worst_mom, mom)
        rec = dict(step=t, ms=ms, finite=finite, momentum_residual=mom, repeated=sim.repeated_steps - rep0,
                   device_allocs=torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - a0,
                   reserved_gib=torch.cuda.memory_stats(dev)["reserved_bytes.all.current"] / 2 ** 30,
                   max_speed=float(state[1].norm(dim=1).max()))
        if before is not None:
            pos_ref, _ = ref.step(before)
            rec["oracle_rel_err"] = float(np.abs(pos.cpu().numpy() - pos_ref).max() / np.abs(pos_ref).max())
            worst_parity = max(worst_parity, rec["oracle_rel_err"])
        recs.append(rec)
        if os.environ.get("ROLLOUT_VERBOSE") and (t % 10 == 0 or rec["repeated"] or rec["device_allocs"]):
            print(json.dumps(rec), flush=True)
        if not finite:
            break
    ms = np.array([r["ms"] for r in recs])
    summary = dict(rollout=args.name, particles=n, boundary=int(state[4].shape[0]), steps=len(recs), all_finite=all(r["finite"] for r in recs),
                   worst_momentum_residual=worst_mom, oracle_steps=min(args.oracle_steps, len(recs)), worst_oracle_rel_err=worst_parity,
                   repeated_steps=int(sum(r["repeated"] for r in recs)), steps_with_device_alloc=int(sum(1 for r in recs[3:] if r["device_allocs"] > 0)),
                   device_allocs_after_step3=int(sum(r["device_allocs"] for r in recs[3:])),
                   device_allocs_total=int(torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - allocs0),
                   ms_median=float(np.median(ms)), ms_max_after_step3=float(ms[3:].max()) if len(ms) > 3 else None,
                   ms_p99=float(np.percentile(ms[3:], 99)) if len(ms) > 3 else None, wall_s=time.time() - t_all,
                   particle_steps_per_s=n / (np.median(ms) * 1e-3), max_speed_last=recs[-1]["max_speed"],
                   reserved_gib_last=recs[-1]["reserved_gib"])
    if args.out:
        json.dump(dict(summary=summary, steps=recs), open(args.out, "w"))
    print(json.dumps(summary), flush=True)


if __name__ == "__main__":
    main()
