"""The reference's ``run_sample.py`` (the canyon demo) on the MI355X path: load a config + checkpoint + one scene
file, roll the scene out with optional particle inflow, write the result.

    python -m dmcf_amd.run_sample -c configs/Liquid3d.yml --ckpt_path checkpoints/Liquid3d/ckpt \\
        --data_path datasets/canyon_data/canyon.msgpack.zst --inflow 200 --timesteps 400 --output_dir output

``run_rollout`` reproduces run_sample.py:142-185: the first frame of the scene is the inflow block; it starts with the
velocity offset (10, 0, -6), acceleration (0, grav, 0), and is appended again after every odd step ``t < inflow``.
Output: ``<output_dir>/example/0000/0000.hdf5`` through ``write_results`` (run_sample.py:223-240) when h5py is
available, otherwise the same arrays as ``0000.npz``.
"""
import argparse
import os
import time

import numpy as np
import torch

INFLOW_VELOCITY = (10.0, 0.0, -6.0)  # run_sample.py:153-155


def initial_state(data, grav, device):
    """run_sample.py:151-162: [pos, vel, acc, None, box, box_normals] of the scene's first frame."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)  # noqa: E731
    in_pos = t(data["pos"])
    in_vel = t(data["vel"]) + torch.tensor([INFLOW_VELOCITY], dtype=torch.float32, device=device)
    in_acc = torch.zeros_like(in_pos) + torch.tensor([[0.0, grav, 0.0]], dtype=torch.float32, device=device)
    return [in_pos, in_vel, in_acc, None, t(data["box"]), t(data["box_normals"])]


def run_rollout(sim, data, timesteps=2, inflow=0):
    """run_sample.py:142-185 -> (list of [N_t, 3] position tensors, seconds per step)."""
    inputs = initial_state(data, sim.model.grav, sim.device)
    in_pos, in_vel, in_acc = inputs[0], inputs[1], inputs[2]
    results = [inputs[0]]
    sim.run_inference([inputs])  # "dummy init" (:165): creates the lazily built weights
    timing = []
    for t in range(timesteps - 1):
        torch.cuda.synchronize(sim.device)
        start = time.time()
        inputs = sim.run_inference([inputs])[0]
        torch.cuda.synchronize(sim.device)
        timing.append(time.time() - start)
        results.append(inputs[0])
        if inflow > t and t % 2 == 1:  # :173-177
            inputs[0] = torch.cat([inputs[0], in_pos], dim=0)
            inputs[1] = torch.cat([inputs[1], in_vel], dim=0)
            inputs[2] = torch.cat([inputs[2], in_acc], dim=0)
    return results, timing


def main(argv=None):
    from . import models, pipelines
    from .datasets import read_scene, write_results, write_results_npz
    from .utils import tf_checkpoint as tc
    from .utils.config import Config
    ap = argparse.ArgumentParser(description="Run a network on one scene (run_sample.py)")
    ap.add_argument("-c", "--cfg_file", required=True)
    ap.add_argument("--ckpt_path", required=True)
    ap.add_argument("--data_path", required=True)
    ap.add_argument("--inflow", default=0, type=int)
    ap.add_argument("--timesteps", default=None, type=int)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--output_dir", default="output")
    args = ap.parse_args(argv)
    cfg = Config.load_from_file(args.cfg_file)
    model = getattr(models, cfg.model.name)(**cfg.model)
    tc.load_into_model(model, tc.load_checkpoint(args.ckpt_path), device=args.device)
    sim = pipelines.Simulator(model, None, device=args.device)
    data = read_scene(args.data_path)
    results, timing = run_rollout(sim, data[0], len(data) if args.timesteps is None else args.timesteps, args.inflow)
    print("Average runtime: %.05f" % (np.mean(timing) if timing else 0.0))
    pos = np.ones((len(results), results[-1].shape[0], 3)) * 1000  # :217-220: absent particles parked at 1000
    for i, r in enumerate(results):
        pos[i, :r.shape[0]] = r.cpu().numpy()
    out_dir = os.path.join(args.output_dir, "example", "0000")
    os.makedirs(out_dir, exist_ok=True)
    output = [(pos, {"name": "pred", "type": "PARTICLE"}), (data[0]["box"], {"name": "bnd", "type": "PARTICLE"})]
    try:
        write_results(os.path.join(out_dir, "%04d.hdf5" % 0), model.name, output)
    except ImportError:
        write_results_npz(os.path.join(out_dir, "%04d.npz" % 0), model.name, output)
    return pos


if __name__ == "__main__":
    main()
