"""Which readings of the absent Open3D operators does the reference's own data separate?

The oracle restates open3d 0.15.2 (not in /root/reference) from its published algorithm; no golden outputs exist
to pin the restatement.  The reference DOES hold physical ground truth: consecutive frames of its canyon scene and
weights trained to continue them (tools/canyon.py).  This tool steps the ORACLE from frames 8 ... 11 under every
disputed reading it can express and prints, per reading, the error ratio (network step vs bare integration, against
the next frame; < 1 = the network helps) and the cosine between the network's correction and the needed one.

A reading is "separated" when it is worse than the shipped one in EVERY frame by >= 10 % of the ratio.  Readings the
data cannot separate stay unpinned until tools/capture_golden.py runs off-box.

    python tools/reading_sweep.py [--md profiles/r06_reading_sweep.md]

Test infrastructure only (imports oracle/); CPU, ~3 min.
"""
import argparse
import contextlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from oracle import model_ref as MR  # noqa: E402
from tools import canyon, configs  # noqa: E402


@contextlib.contextmanager
def patched(obj, name, value):
    old = getattr(obj, name)
    setattr(obj, name, value)
    try:
        yield
    finally:
        setattr(obj, name, old)


def filters_through(fn):
    """Every filter array handed to the operator (after the ASCC mirror, convolutions.py:410-412) goes through fn."""
    orig = O.continuous_conv

    def conv(filters, *a, **k):
        return orig(np.ascontiguousarray(fn(np.asarray(filters))), *a, **k)
    return patched(O, "continuous_conv", conv)


def conv_kwargs(**over):
    """ContinuousConvRef built with other operator attributes (align_corners, interpolation ...)."""
    orig = O.ContinuousConvRef

    def make(*a, **k):
        k.update(over)
        return orig(*a, **k)
    return patched(O, "ContinuousConvRef", make)


def window_of(fn):
    orig = O.window
    return patched(O, "window", lambda typ, q, fac=1.0: fn(orig, typ, q, fac))


def mirror_as(fn):
    return patched(O, "mirror_kernel", fn)


def extent_scaled(s):
    """The filter extent handed to the layers = s x 2 x particle_radii (pbf_model.py:328 has s = 1)."""
    orig = MR.ModelRef._cconv

    def cc(self, index, alias, feats, inp_pos, out_pos, extent, *a, **k):
        return orig(self, index, alias, feats, inp_pos, out_pos, np.float32(extent) * np.float32(s), *a, **k)
    return patched(MR.ModelRef, "_cconv", cc)


def reading_list():
    """(group, name, cfg overrides, context manager factory)."""
    none = contextlib.nullcontext
    R = [("shipped", "the oracle as committed", {}, none)]
    # --- filter layout [D, H, W] = (z, y, x) and orientation
    R += [("filter axes", "x <-> z", {}, lambda: filters_through(lambda f: f.transpose(2, 1, 0, 3, 4))),
          ("filter axes", "x <-> y", {}, lambda: filters_through(lambda f: f.transpose(0, 2, 1, 3, 4))),
          ("filter axes", "y <-> z", {}, lambda: filters_through(lambda f: f.transpose(1, 0, 2, 3, 4))),
          ("filter orientation", "all three axes flipped", {}, lambda: filters_through(lambda f: f[::-1, ::-1, ::-1])),
          ("filter orientation", "x flipped", {}, lambda: filters_through(lambda f: f[:, :, ::-1])),
          ("filter orientation", "y flipped", {}, lambda: filters_through(lambda f: f[:, ::-1])),
          ("filter orientation", "z flipped", {}, lambda: filters_through(lambda f: f[::-1]))]
    # --- interpolation, align_corners / the (size - 1) scaling, coordinate map
    R += [("interpolation", "nearest_neighbor", dict(interpolation="nearest_neighbor"), none),
          ("interpolation", "linear_border", dict(interpolation="linear_border"), none),
          ("align_corners", "False: x*size - 0.5 instead of x*(size-1)", {}, lambda: conv_kwargs(align_corners=False)),
          ("coordinate map", "ball_to_cube_radial", dict(coordinate_mapping="ball_to_cube_radial"), none),
          ("coordinate map", "identity", dict(coordinate_mapping="identity"), none)]
    # --- window: argument and function
    R += [("window", "none", dict(window=None, window_sym=None), none),
          ("window", "argument d/R instead of d^2/R^2", {}, lambda: window_of(lambda w, t, q, f: w(t, np.sqrt(q), f))),
          ("window", "argument d^4/R^4", {}, lambda: window_of(lambda w, t, q, f: w(t, q * q, f))),
          ("window", "poly6 <-> peak swapped", dict(window="peak", window_sym="poly6"), none),
          ("window", "ASCC window poly6", dict(window_sym="poly6"), none)]
    # --- ASCC mirror
    R += [("ASCC mirror", "sym_axis 0", dict(sym_axis=0), none),
          ("ASCC mirror", "sym_axis 2", dict(sym_axis=2), none),
          ("ASCC mirror", "+k[::-1,::-1,::-1] (symmetric instead of antisymmetric)", {},
           lambda: mirror_as(lambda k, a: np.concatenate([np.asarray(k)[::-1, ::-1, ::-1], np.asarray(k)], axis=a))),
          ("ASCC mirror", "halves in the other order: concat([k, -k[::-1,...]])", {},
           lambda: mirror_as(lambda k, a: np.concatenate([np.asarray(k), -np.asarray(k)[::-1, ::-1, ::-1]], axis=a))),
          ("ASCC mirror", "flip along sym_axis only", {},
           lambda: mirror_as(lambda k, a: np.concatenate([-np.flip(np.asarray(k), axis=a), np.asarray(k)], axis=a)))]
    # --- scales
    R += [("out_scale", "x 0.5", dict(out_scale=[0.00390625] * 3), none),
          ("out_scale", "x 2", dict(out_scale=[0.015625] * 3), none),
          ("filter extent", "= radius (x 0.5)", {}, lambda: extent_scaled(0.5)),
          ("filter extent", "x 1.25", {}, lambda: extent_scaled(1.25))]
    # --- grid_pos, search set
    R += [("grid_pos", "centralize off", dict(centralize=False), none),
          ("grid_pos", "hysteresis 0", dict(sample_hyst=0.0), none),
          ("grid_pos", "hysteresis 0.25", dict(sample_hyst=0.25), none),
          ("grid_pos", "pad 1", dict(sample_pad=1), none),
          ("search set", "all 27 voxels (the distance test)", {}, lambda: O.search_bins("all")),
          ("search set", "8 corner voxels", {}, lambda: O.search_bins("corners"))]
    return R


def run(frames=canyon.FRAMES, log=print):
    fx = canyon.load()
    w = dict(np.load(canyon.WEIGHTS))
    rows = []
    for group, name, over, ctx in reading_list():
        cfg = dict(configs.LIQUID3D)
        cfg.update(over)
        t0 = time.time()
        with ctx():
            ref = MR.ModelRef(cfg, w)
            sc = []
            for t in frames:
                try:
                    pos, _ = ref.step(canyon.inputs(fx, t))
                    sc.append(canyon.score(fx, t, pos))
                except Exception as e:  # a reading the restatement cannot evaluate (shape mismatch) is reported, not hidden
                    sc.append((float("nan"), float("nan")))
                    log(f"  {group} / {name}: frame {t}: {type(e).__name__}: {e}")
            try:  # the sharper figure: four free-running steps from frame 8 against frame 12 (tools/canyon.rollout_score)
                roll = canyon.rollout_score(fx, lambda p, v: ref.step([p, v, None, None, fx["box"], fx["box_normals"]]))
            except Exception as e:
                roll = float("nan")
                log(f"  {group} / {name}: rollout: {type(e).__name__}: {e}")
        rows.append((group, name, sc, roll))
        log(f"{group:20s} {name:60s} ratio " + " ".join(f"{r:8.3f}" for r, _ in sc) + "  cos "
            + " ".join(f"{c:6.2f}" for _, c in sc) + f"  rollout {roll:.3f}  ({time.time() - t0:.0f} s)")
    return rows


def verdicts(rows):
    base, base_roll = rows[0][2], rows[0][3]
    out = []
    for group, name, sc, roll in rows:
        if group == "shipped":
            out.append("—")
            continue
        if any(np.isnan(r) for r, _ in sc):
            out.append("not evaluable")
            continue
        worse = all(r >= 1.1 * b for (r, _), (b, _) in zip(sc, base))
        better = all(r <= b / 1.1 for (r, _), (b, _) in zip(sc, base))
        if worse:
            out.append("**separated** (worse in every frame)")
        elif roll >= 1.1 * base_roll:
            out.append("**separated by the rollout** (one step cannot tell)")
        elif better and roll <= base_roll / 1.1:
            out.append("BETTER than shipped in every frame and in the rollout")
        else:
            out.append("not separated")
    return out


def to_markdown(rows, frames):
    v = verdicts(rows)
    lines = ["| group | reading | " + " | ".join(f"ratio t={t}" for t in frames) + " | mean ratio | mean cosine | 4-step rollout ratio | verdict |",
             "|---|---|" + "---:|" * (len(frames) + 3) + "---|"]
    for (group, name, sc, roll), verdict in zip(rows, v):
        r = [x for x, _ in sc]
        c = [x for _, x in sc]
        lines.append(f"| {group} | {name} | " + " | ".join(f"{x:.3f}" for x in r)
                     + f" | {np.mean(r):.3f} | {np.mean(c):.2f} | {roll:.3f} | {verdict} |")
    return "\n".join(lines)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--md", default=None, help="write the table here")
    args = ap.parse_args()
    O.build()
    rows = run()
    md = to_markdown(rows, canyon.FRAMES)
    print(md)
    if args.md:
        with open(args.md, "w") as f:
            f.write("# Reading sweep: the oracle against the reference's canyon frames (round 6)\n\n"
                    "`python tools/reading_sweep.py --md " + args.md + "` -- CPU, the ORACLE only.  One step of the Liquid3d "
                    "SymNet (reference checkpoint) from frame t of `datasets/canyon_data/canyon.msgpack.zst` against frame "
                    "t + 1.  ratio = mean |step - next frame| / mean |integration only - next frame| (< 1: the network "
                    "helps); cosine between the network's correction and the needed one.  A reading is *separated* when "
                    "its ratio is >= 1.1 x the shipped reading's in every frame, *separated by the rollout* when only the 4-step free-running "
                    "rollout from frame 8 (against frame 12, same ratio) is >= 1.1 x the shipped one.\n\n" + md + "\n")
