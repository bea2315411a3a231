"""Diagnostic: every CConv call of one oracle model step replayed on the HIP kernel with identical inputs."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O
from oracle.model_ref import ModelRef
from dmcf_amd import ops
from tools import configs, scenes

dev = torch.device("cuda:0")
w = dict(np.load(os.path.join(ROOT, "tests/golden/liquid3d_weights.npz")))
cfg = configs.LIQUID3D
r32 = ModelRef(cfg, w)
r32.record = []
r32.step(scenes.model_inputs(scenes.box_scene(12)))
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for rec in r32.record:
    conv64 = O.ContinuousConvRef(rec["kernel"], bias=rec["bias"], window_function=rec["window"], ignore_query_points=rec["ignore"],
                                 symmetric=rec["symmetric"], sym_axis=1, f64=True)
    ref64 = conv64(rec["feats"], rec["inp_pos"], rec["out_pos"], np.float32(rec["extent"]), nns=rec["nns"])
    idx, rs, d = rec["nns"]
    y = ops.cconv_forward(T(rec["kernel"]), T(rec["out_pos"]), rec["extent"], T(rec["inp_pos"]), T(rec["feats"]), T(idx), T(rs),
                          neighbors_value=T(d), window=rec["window"], symmetric=rec["symmetric"], sym_axis=1,
                          bias=None if rec["bias"] is None else T(rec["bias"])).cpu().numpy()
    nn = ops.fixed_radius_search(T(rec["inp_pos"]), T(rec["out_pos"]), 0.5 * rec["extent"], ignore_query_point=rec["ignore"])
    same = np.array_equal(nn.neighbors_row_splits.cpu().numpy(), rs)
    sc = np.abs(ref64).max()
    print("conv %2d  %s cin %2d cout %2d  pairs %8d  hip-vs-64 %.2e  ref32-vs-64 %.2e  rows_equal %s" % (
        rec["index"], "ASCC" if rec["symmetric"] else "    ", rec["kernel"].shape[3], rec["kernel"].shape[4], idx.size,
        np.abs(y - ref64).max() / sc, np.abs(rec["out"] - ref64).max() / sc, same))
