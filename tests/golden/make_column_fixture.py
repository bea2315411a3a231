"""Generates tests/golden/column_test.npz: the two TEST scenes of configs/column/hrnet.yml (dataset section: type column,
res 100, gravity -10.0, dt 0.0025; test: seed 44, offset 10.0, pts_cnt [1, 5], data_cnt 2, timesteps 200), produced by the
REFERENCE's own generator -- datasets/column_gen.py, numpy only -- called exactly as DatasetGroup.gen_data does
(datasets/dataset_reader_physics.py:144-165: np.random.seed(seed), then gen_data(**cfg)).

Run in the build container (imports /root/reference/datasets/column_gen.py by path); the GPU box only sees the committed
.npz.  What is stored is DATA -- per scene the [T, n, 3] fluid positions / velocities of the reference's 1-D SPH solver, the
boundary points and normals, the per-frame gravity -- not reference source.  Used as the inputs of BASELINE.json config 1
(tests/test_gpu_model.py::test_column_config1_on_reference_generated_scenes) and by the oracle-side CPU test."""
import importlib.util
import os

import numpy as np

REF = "/root/reference/datasets/column_gen.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "column_test.npz")

if __name__ == "__main__":
    spec = importlib.util.spec_from_file_location("ref_column_gen", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cfg = dict(offset=10.0, pts_cnt=[1, 5], data_cnt=2, timesteps=200, res=100, gravity=-10.0, dt=0.0025)  # test + dataset keys
    np.random.seed(44)
    data = mod.gen_data(**cfg)
    out = {}
    for s, scene in enumerate(data):
        out[f"s{s}_pos"] = np.stack([np.asarray(f["pos"], np.float32) for f in scene])
        out[f"s{s}_vel"] = np.stack([np.asarray(f["vel"], np.float32) for f in scene])
        out[f"s{s}_grav"] = np.stack([np.asarray(f["grav"], np.float32) for f in scene])
        out[f"s{s}_box"] = np.asarray(scene[0]["box"], np.float32)
        out[f"s{s}_box_normals"] = np.asarray(scene[0]["box_normals"], np.float32)
        print(s, {k: (np.asarray(v).shape if not np.isscalar(v) else v) for k, v in scene[0].items()})
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})
