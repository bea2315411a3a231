"""Where a step of the small configurations (BASELINE.json configs 2 / 3 / 4) spends its HOST time: cProfile over steady-state
steps + the device -> host synchronisations torch reports (set_sync_debug_mode).
    python tools/profile_small.py waterramps 200"""
import cProfile, io, os, pstats, sys, time, warnings
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import long_rollout, scenes


def main():
    name, steps = sys.argv[1], int(sys.argv[2])
    from dmcf_amd import models
    from dmcf_amd.pipelines import Simulator
    from dmcf_amd.pipelines.simulator import steady_steps
    from dmcf_amd.utils import tf_checkpoint as tc
    dev = torch.device("cuda:0")
    cfg, w, scene, grav = long_rollout.setup(name)
    model = getattr(models, cfg["name"])(**cfg)
    tc.load_into_model(model, w, device=dev)
    sim = Simulator(model, device="cuda", reserve_gib="auto")
    state = scenes.model_inputs(scene, device=dev, grav=grav)
    for _ in range(20):
        state = sim.step([state])[0]
    torch.cuda.synchronize()
    with steady_steps():
        t0 = time.perf_counter()
        for _ in range(steps):
            state = sim.step([state])[0]
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        print(f"{name}: {state[0].shape[0]} particles, {ms:.3f} ms per step (eager, steady state)")
        # host-only cost: the same loop without waiting for the GPU at the end of each step is what the above is; now profile
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(steps):
            state = sim.step([state])[0]
        torch.cuda.synchronize()
        pr.disable()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
        print("\n".join(l[:150] for l in s.getvalue().splitlines()[:60]))
        # synchronisations
        torch.cuda.set_sync_debug_mode("warn")
        with warnings.catch_warnings(record=True) as ws:
            warnings.simplefilter("always")
            state = sim.step([state])[0]
        torch.cuda.set_sync_debug_mode("default")
        print(f"synchronising calls in one step: {len(ws)}")
        seen = {}
        for wv in ws:
            k = f"{os.path.basename(wv.filename)}:{wv.lineno}"
            seen[k] = seen.get(k, 0) + 1
        print(seen)


if __name__ == "__main__":
    main()
