"""Diagnostic: per-step allocator behaviour of a rollout -- peak allocated bytes, reserved bytes, device mallocs, the longest
row of every search class, point-set sizes -- printed for the steps that are slow or allocate.
usage: python tools/diag_rollout_mem.py liquid3d_dam 200"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import long_rollout, scenes  # noqa: E402


def main():
    name, steps = sys.argv[1], int(sys.argv[2])
    from dmcf_amd import models, ops
    from dmcf_amd.pipelines import Simulator
    from dmcf_amd.utils import tf_checkpoint as tc
    from dmcf_amd.utils.convolutions import _CACHE
    dev = torch.device("cuda:0")
    cfg, w, scene, grav = long_rollout.setup(name)
    model = getattr(models, cfg["name"])(**cfg)
    tc.load_into_model(model, w, device=dev)
    sim = Simulator(model, device="cuda", reserve_gib="auto")
    state = scenes.model_inputs(scene, device=dev, grav=grav)
    med = []
    for t in range(steps):
        torch.cuda.reset_peak_memory_stats(dev)
        a0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
        rep0 = sim.repeated_steps
        ops.timer = ops.LaunchTimer() if os.environ.get("DIAG_TIMER") else None
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        state = sim.step([state])[0]
        torch.cuda.synchronize(dev)
        ms = 1e3 * (time.perf_counter() - t0)
        st = torch.cuda.memory_stats(dev)
        allocs = st.get("num_device_alloc", 0) - a0
        med.append(ms)
        slow = t > 3 and ms > 1.5 * float(np.median(med[-50:]))
        if t < 2 or allocs or slow or sim.repeated_steps > rep0 or t % 50 == 0:
            hints = {k: v for k, v in sorted(_CACHE.hints.items(), key=lambda kv: str(kv[0]))}
            sets = [int(p.shape[0]) for p in getattr(model, "dilated_pos", [])]
            pos = state[0]
            print(f"step {t}: {ms:.1f} ms, repeated {sim.repeated_steps - rep0}, device mallocs {allocs}, peak allocated "
                  f"{st['allocated_bytes.all.peak'] / 2**30:.2f} GiB, reserved {st['reserved_bytes.all.current'] / 2**30:.2f} GiB, "
                  f"largest request peak {st.get('requested_bytes.all.peak', 0) / 2**30:.2f}; point sets {sets}; bbox "
                  f"{[round(float(v), 1) for v in pos.min(0).values.tolist() + pos.max(0).values.tolist()]}; longest rows "
                  f"{sorted(set(int(v) for v in hints.values()))}", flush=True)
            if sim.repeated_steps > rep0:
                print("   outgrown:", _CACHE.last_overflow, flush=True)
            if ops.timer is not None:
                by = {}
                for k, m, x in ops.timer.results():
                    by[k] = by.get(k, 0.0) + x
                print("   kernels ms:", {k: round(v, 1) for k, v in by.items()}, flush=True)


if __name__ == "__main__":
    main()
