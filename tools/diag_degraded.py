"""Diagnostic: the 1M bench scene after N rollout steps on the GPU (particles leaking through the shell),
then ONE oracle step from that state with every CConv call replayed on the HIP kernels with identical inputs, the neighbour
search compared row by row, and the lattices compared point by point.  usage: python tools/diag_degraded.py [side] [steps]
SCENE=liquid3d_dam: the dam break of config 4 instead of the box; DMCF_FRS_SET=open3d: that neighbour set on both sides."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.model_ref import ModelRef
from dmcf_amd import models, ops
from dmcf_amd.pipelines import Simulator
from dmcf_amd.utils import tf_checkpoint as tc
from dmcf_amd.utils.tools.losses import grid_pos
from tools import configs, scenes

side = int(sys.argv[1]) if len(sys.argv) > 1 else 100
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
dev = torch.device("cuda:0")
w = dict(np.load(os.path.join(ROOT, "tests/golden/liquid3d_weights.npz")))
cfg = configs.LIQUID3D
model = getattr(models, cfg["name"])(**cfg)
tc.load_into_model(model, w, device=dev)
sim = Simulator(model, device="cuda", reserve_gib="auto")
if os.environ.get("SCENE"):
    from tools import long_rollout
    state = scenes.model_inputs(long_rollout.setup(os.environ["SCENE"])[2], device=dev)
else:
    state = scenes.model_inputs(scenes.box_scene(side), device=dev)
import oracle as _oracle
_bins = _oracle.search_bins({"distance": "all", "open3d": "own+corners", "open3d_corners": "corners"}[os.environ.get("DMCF_FRS_SET", "distance")])
_bins.__enter__()
for _ in range(steps):
    state = sim.step([state])[0]
before = [None if x is None else x.cpu().numpy() for x in state]
out = sim.step([state])[0]
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


_seen = set()


def replay(rec):
    idx, rs, d = rec["nns"]
    R = 0.5 * rec["extent"]
    inp, outp = T(rec["inp_pos"]), T(rec["out_pos"])
    nn = ops.fixed_radius_search(inp, outp, R, ignore_query_point=rec["ignore"], return_distances=False)
    rs_h = nn.neighbors_row_splits.cpu().numpy()
    rows_equal = np.array_equal(rs_h, rs)
    sets_equal = None
    if rows_equal:  # same rows as sets: compare the per-row index sums and sums of squares
        i64 = nn.neighbors_index.long()
        seg = torch.repeat_interleave(torch.arange(outp.shape[0], device=dev), torch.diff(nn.neighbors_row_splits))
        a1 = torch.zeros(outp.shape[0], dtype=torch.int64, device=dev).index_add_(0, seg, i64)
        a2 = torch.zeros(outp.shape[0], dtype=torch.int64, device=dev).index_add_(0, seg, i64 * i64 % 1000003)
        j = T(idx).long()
        b1 = torch.zeros_like(a1).index_add_(0, seg, j)
        b2 = torch.zeros_like(a2).index_add_(0, seg, j * j % 1000003)
        sets_equal = bool(torch.equal(a1, b1) and torch.equal(a2, b2))
    if not rows_equal and id(rec["nns"]) not in _seen:
        _seen.add(id(rec["nns"]))
        bad = np.nonzero(np.diff(rs_h) != np.diff(rs))[0]
        hidx = nn.neighbors_index.cpu().numpy()
        for r in bad[:4]:
            a, b = set(hidx[rs_h[r]:rs_h[r + 1]].tolist()), set(idx[rs[r]:rs[r + 1]].tolist())
            q = rec["out_pos"][r]
            for j in sorted(a ^ b):
                pj = rec["inp_pos"][j]
                d32 = np.float32(np.float32(np.float32(pj[0] - q[0]) ** 2 + np.float32(pj[1] - q[1]) ** 2) + np.float32(pj[2] - q[2]) ** 2)
                d64 = float(((pj.astype(np.float64) - q.astype(np.float64)) ** 2).sum())
                print("   row %d (query %s): pair with %d (%s) only in %s: d2 f32 %.9g  f64 %.12g  R2 f32 %.9g  |d|/R %.6f" % (
                    r, q, j, pj, "HIP" if j in a else "oracle", d32, d64, np.float32(R) * np.float32(R), d64 ** 0.5 / R), flush=True)
    cin, cout = rec["kernel"].shape[3], rec["kernel"].shape[4]
    res = []
    for hint in ((0, 2) if cin > 16 and not rec["symmetric"] else (0,)):
        y = ops.cconv_forward(T(rec["kernel"]), outp, rec["extent"], inp, T(rec["feats"]), T(idx), T(rs), neighbors_value=T(d),
                              window=rec["window"], symmetric=rec["symmetric"], sym_axis=1,
                              bias=None if rec["bias"] is None else T(rec["bias"]), row_length_hint=hint).cpu().numpy()
        res.append("hint %d: %.2e" % (hint, rel(y, rec["out"])))
    print("conv %2d %s %2d -> %2d  R %.2f  n_in %8d n_out %8d pairs %10d longest row %5d | rows %s sets %s | hip vs oracle32 %s" % (
        rec["index"], "ASCC" if rec["symmetric"] else "    ", cin, cout, R, inp.shape[0], outp.shape[0], idx.size, int(np.diff(rs).max()),
        rows_equal if rows_equal else "DIFFER (%d rows)" % int((np.diff(rs_h) != np.diff(rs)).sum()), sets_equal, ", ".join(res)), flush=True)


ref = ModelRef(cfg, w)
ref.record = replay
pos_ref, vel_ref = ref.step(before)
print("step %d: pos hip-vs-oracle %.2e  correction %.2e" % (steps + 1, rel(out[0].cpu().numpy(), pos_ref),
                                                           rel(model.pos_correction.cpu().numpy(), ref.pos_correction)))
_e = np.abs(out[0].cpu().numpy() - pos_ref).max(axis=1)
_w = np.argsort(-_e)[:5]
print("worst particles:", [(int(i), float(_e[i]), before[0][i].tolist(), float(np.linalg.norm(before[1][i])),
                            model.pos_correction[i].cpu().numpy().tolist(), ref.pos_correction[i].tolist()) for i in _w])
# the lattices
allp = np.concatenate([ref.pos_adv, ref.box_kept]) if hasattr(ref, "pos_adv") else None
for k, s in enumerate(getattr(ref, "dilated_pos", []) or []):
    h = model.dilated_pos[k].cpu().numpy()
    print("scale %d: oracle %d points, hip %d points, equal %s" % (k, s.shape[0], h.shape[0], s.shape == h.shape and np.array_equal(s, h)))
