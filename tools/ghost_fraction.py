"""Ghost fractions of the block-sharded rollout, measured with VIRTUAL ranks on one MI355X: the ranks
are threads of one process (dmcf_amd.parallel.LocalComm), every one runs the real HIP kernels on its own block of ONE box of
(gx*side) x (gy*side) x (gz*side) fluid particles -- the scene pieces `bench.py --gpus N` gives its ranks.  Prints, per rank
and per point set, owned points, ghosts of the set's widest plan and of the per-layer plans derived from it, and the rows /
bytes of features exchanged per step, and the wall time of a step of all ranks divided by their number (the GPU time a
rank's step costs, ghosts and exchanges included -- the ranks' kernels serialise on the one device).
usage: python tools/ghost_fraction.py [side=100] [gx gy gz = 2 2 2] [steps=2] [weak | strong]
``strong``: ONE box of side^3 fluid particles split over the gx * gy * gz ranks (blocks of side / g cells per axis: BASELINE.json's
"1M particles @ 1/2/4/8 GPUs" read literally); ``weak`` (default): side^3 particles PER rank."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmcf_amd import models, parallel  # noqa: E402
from dmcf_amd.utils import tf_checkpoint as tc  # noqa: E402
from tools import configs, scenes  # noqa: E402


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    grid = [int(x) for x in sys.argv[2:5]] if len(sys.argv) > 4 else [2, 2, 2]
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 2
    strong = len(sys.argv) > 6 and sys.argv[6] == "strong"
    world = grid[0] * grid[1] * grid[2]
    if strong:
        assert all(side % g == 0 for g in grid), "strong: the box's side must be divisible by the ranks per axis"
    block = [side // g for g in grid] if strong else [side] * 3  # lattice cells of a rank's block per axis
    torch.cuda.init()  # (before the rank threads make their first device calls concurrently)
    dev = torch.device("cuda:0")
    h = 0.05
    decomp = parallel.BlockDecomposition.uniform([0.0, 0.0, 0.0], [g * b * h for g, b in zip(grid, block)], grid)
    weights = dict(np.load(os.path.join(ROOT, "tests", "golden", "liquid3d_weights.npz")))

    def rank_fn(comm):
        cfg = configs.LIQUID3D
        model = getattr(models, cfg["name"])(**cfg)
        tc.load_into_model(model, weights, device=dev)
        sim = parallel.ShardedSimulator(model, comm, decomp)
        state = parallel.shard_scene(scenes.box_block_scene(block, grid, comm.rank), decomp, comm.rank, dev, presharded=True)
        rows, times = [], []
        for _ in range(steps):
            before = sim.exchanged_rows
            torch.cuda.synchronize(dev)
            comm.hub.barrier.wait()
            t0 = time.perf_counter()
            state = sim.step(state)
            torch.cuda.synchronize(dev)
            comm.hub.barrier.wait()
            times.append(time.perf_counter() - t0)
            rows.append(sim.exchanged_rows - before)
        sets = {}
        for name, pos in sim._sets.items():
            plans = {f"{w:g}": int(p.ghost_pos.shape[0]) for (n, w), p in sorted(sim._plans.items()) if n == name}
            sets[str(name)] = dict(owned=int(pos.shape[0]), widest=float(sim._wide[name].width),
                                   ghosts_widest=int(sim._wide[name].ghost_pos.shape[0]), ghosts_by_width=plans)
        launch = {}
        for name, r, n_rows, own in sim.launch_rows:  # input rows of every convolution launch: owned + the ghosts within ITS radius
            k = f"{name}@{r:g}"
            launch[k] = dict(launches=launch.get(k, {}).get("launches", 0) + 1, owned=own, ghosts=n_rows - own)
        return dict(rank=comm.rank, block=decomp.coords(comm.rank), fluid=int(state["pos"].shape[0]), sets=sets,
                    feature_rows_per_step=rows[-1], step_seconds=times, host_syncs_per_step=sim.host_syncs_last_step,
                    launch_inputs=launch)

    res = parallel.run_local_ranks(world, rank_fn)
    out = dict(side=side, grid=grid, scaling="strong" if strong else "weak", fluid_total=block[0] * block[1] * block[2] * world, ranks=res)
    print(json.dumps(out))
    last = max(r["step_seconds"][-1] for r in res)
    per_step = [max(r["step_seconds"][i] for r in res) for i in range(steps)]
    print(f"steps (all {world} ranks on ONE GPU, ms): " + ", ".join(f"{1e3 * t:.1f}" for t in per_step) + f"; per rank: "
          + ", ".join(f"{1e3 * t / world:.1f}" for t in per_step) + f"; host round trips per step and rank: {res[0]['host_syncs_per_step']}",
          file=sys.stderr)
    print(f"last step: {1e3 * last:.1f} ms for all {world} ranks on ONE GPU = {1e3 * last / world:.1f} ms of GPU time per rank and step "
          "(the ranks' kernels share the device; their host work overlaps)", file=sys.stderr)
    r0 = res[0]
    print("rank 0 launch inputs (owned + ghosts within the launch's own radius): "
          + ", ".join(f"{k}: {v['launches']} x ({v['owned']} + {v['ghosts']})" for k, v in sorted(r0["launch_inputs"].items())), file=sys.stderr)
    # summary on stderr
    for r in res:
        s = r["sets"]
        line = ", ".join(f"{k}: {v['owned']} owned + {v['ghosts_widest']} ghosts ({100.0 * v['ghosts_widest'] / max(v['owned'], 1):.1f} % at width {v['widest']:g})"
                         for k, v in s.items())
        print(f"rank {r['rank']} block {r['block']}: {line}; {r['feature_rows_per_step']} feature rows received per step", file=sys.stderr)


if __name__ == "__main__":
    main()
