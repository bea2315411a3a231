"""Per-step wall times of the 1M bench workload (diagnostic for allocator / host-sync effects)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmcf_amd import models, ops
from dmcf_amd.pipelines import Simulator
from dmcf_amd.utils import tf_checkpoint as tc
from tools import configs, scenes
side = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
cfg = configs.LIQUID3D
model = getattr(models, cfg["name"])(**cfg)
tc.load_into_model(model, dict(np.load(os.path.join(ROOT, "tests/golden/liquid3d_weights.npz"))), device=dev)
sim = Simulator(model, device="cuda:0", reserve_gib="auto")
state = scenes.model_inputs(scenes.box_scene(side), device=dev)
for s in range(7):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ops.timer = ops.LaunchTimer()
    state = sim.step([state])[0]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    recs = ops.timer.results(); ops.timer = None
    tot = {}
    for k, m, ms in recs: tot[k] = tot.get(k, 0) + ms
    st = torch.cuda.memory_stats()
    print(f"step {s}: {dt*1e3:7.1f} ms  " + " ".join(f"{k}={v:.0f}" for k, v in tot.items()) +
          f"  reserved={torch.cuda.memory_reserved()/2**30:.1f}GiB allocs={st['num_device_alloc']} frees={st['num_device_free']} retries={st['num_alloc_retries']}")
