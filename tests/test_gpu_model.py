"""GPU parity of the whole per-step path: model(inputs, training=False) on MI355X against the numpy/C
oracle restatement of the reference's model code (oracle/model_ref.py), same seeded inputs.

Bar (BASELINE.json north_star): positions / velocities within 1e-5 relative per step."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but torch.cuda.is_available() is False")
    return torch.device("cuda:0")


def _build(cfg, weights, dev):
    from dmcf_amd import models
    from dmcf_amd.utils import tf_checkpoint as tc
    model = getattr(models, cfg["name"])(**cfg)
    tc.load_into_model(model, weights, device=dev)
    return model


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _displacement_check(pos_before, pos, pos_ref, what, tol=1e-5, ulps=3, noise=0.0):
    """The bar of north_star ("positions within 1e-5 relative per step") ELEMENT-WISE, on what a step changes: per particle
    |dpos_hip - dpos_ref| <= tol * |dpos_ref|, with the floor float32 itself sets -- a position is stored to an ulp of its
    largest coordinate, and two float32 evaluations of pos + dt v + correction may round ``ulps`` of those apart (the same floor
    separates the float32 oracle from the float64-operator oracle).  ``noise``: the absolute float32 noise of the NETWORK OUTPUT
    where the caller has measured it (three times the distance of the float32 oracle's correction from the float64-operator
    oracle's) -- it matters only where corrections are huge (the dissolved bench scene).  The whole-array number (_rel: max error over the largest
    coordinate of the scene) is printed beside it; VERDICT r05 item 6.  Returns (worst err / bar, share of particles whose bar
    is the relative one rather than the ulp floor)."""
    p0 = np.asarray(pos_before, dtype=np.float64)
    d_ref = np.asarray(pos_ref, dtype=np.float64) - p0
    d_hip = np.asarray(pos, dtype=np.float64) - p0
    err = np.abs(d_hip - d_ref).max(axis=1)
    size = np.linalg.norm(d_ref, axis=1)
    floor = ulps * np.spacing(np.abs(np.asarray(pos_ref, dtype=np.float32)).max(axis=1)).astype(np.float64)
    bar = np.maximum(tol * size, floor + noise)
    worst = float((err / bar).max()) if len(err) else 0.0
    share = float((tol * size >= floor).mean()) if len(err) else 0.0
    i = int(np.argmax(err / bar)) if len(err) else 0
    print(f"{what}: element-wise displacement err/bar {worst:.2f} (particle {i}: err {err[i]:.2e}, |dpos| {size[i]:.2e}, "
          f"bar {bar[i]:.2e}; relative bar binds for {100 * share:.1f} % of the particles), whole-array pos {_rel(pos, pos_ref):.2e}")
    assert worst <= 1.0, f"{what}: particle {i} moved {err[i]:.2e} away from the reference displacement (bar {bar[i]:.2e})"
    return worst, share


# The neighbour set a step runs under (dmcf_amd.ops.SEARCH_SETS) and the oracle's statement of the SAME set (oracle.BINS):
# the product's default -- the set of the distance test -- is what the oracle returns when it walks all 27 voxels; the
# emulation of open3d's float walk is the oracle's own default (own voxel + 8 corners).  A capture of the real library will be
# matched under "open3d"; both are held to the same bars here, like against like.
FRS_MODES = {"distance": "all", "open3d": "own+corners"}


class _frs_mode:
    """``with _frs_mode("open3d"): ...`` -- the HIP path under DMCF_FRS_SET=<mode>, every oracle search of the block over the
    matching bins."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        import oracle
        self.saved = os.environ.get("DMCF_FRS_SET")
        os.environ["DMCF_FRS_SET"] = self.mode
        self.bins = oracle.search_bins(FRS_MODES[self.mode])
        self.bins.__enter__()
        return self

    def __exit__(self, *exc):
        self.bins.__exit__(*exc)
        if self.saved is None:
            os.environ.pop("DMCF_FRS_SET", None)
        else:
            os.environ["DMCF_FRS_SET"] = self.saved
        return False


def _compare_step(cfg, weights, scene, dev, grav=None, steps=1, tol=1e-5, mode="distance"):
    with _frs_mode(mode):
        return _compare_step_in_mode(cfg, weights, scene, dev, grav, steps, tol)


def _compare_step_in_mode(cfg, weights, scene, dev, grav, steps, tol):
    from oracle.model_ref import ModelRef
    from dmcf_amd.pipelines import Simulator
    from tools import scenes
    model = _build(cfg, weights, dev)
    sim = Simulator(model, device="cuda")
    ref = ModelRef(cfg, weights)
    ref64 = ModelRef(cfg, weights, f64=True)  # same restatement with the operators evaluated in float64
    data_np = scenes.model_inputs(scene, grav=grav)
    data_t = scenes.model_inputs(scene, device=dev, grav=grav)
    for s in range(steps):
        pos_ref, vel_ref = ref.step(data_np)
        out = sim.step([data_t])[0]
        pos, vel = out[0].cpu().numpy(), out[1].cpu().numpy()
        assert pos.shape == pos_ref.shape
        assert _rel(pos, pos_ref) <= tol, f"step {s}: pos rel err {_rel(pos, pos_ref):.2e}"
        # velocities are (pos' - pos)/dt: one ulp of a position is already ~1e-7*|x|/dt of velocity, so for slow
        # scenes float32 itself cannot hold 1e-5.  The bar is 1e-5, or three times the distance of the float32
        # ORACLE from the float64-operator oracle when that float32 noise floor is above 1e-5.
        pos64, vel64 = ref64.step(data_np)
        floor = _rel(vel_ref, vel64)
        # ... and never tighter than a 4-ulp difference of a position divided by dt (vel' = (pos' - pos)/dt exactly)
        ulp_floor = 4 * np.finfo(np.float32).eps * np.abs(pos64).max() / cfg["timestep"] / np.abs(vel64).max()
        vtol = max(tol, 3 * floor, ulp_floor)
        assert _rel(vel, vel64) <= vtol, f"step {s}: vel rel err {_rel(vel, vel64):.2e} (bar {vtol:.1e}, f32 floor {floor:.1e})"
        _check_correction(model, ref, ref64, f"step {s}")
        _displacement_check(data_np[0], pos, pos_ref, f"step {s}", tol,
                            noise=3 * float(np.abs(ref.pos_correction - ref64.pos_correction).max()))
        data_np = [pos_ref, vel_ref] + data_np[2:]
        data_t = [torch.from_numpy(pos_ref).to(dev), torch.from_numpy(vel_ref).to(dev)] + list(data_t[2:])
    return model, ref


def _check_correction(model, ref, ref64, what):
    """The network OUTPUT (the learned position correction, ~1e-3 of the positions, so invisible in the 1e-5 position bar):
    held to three times the distance of the float32 oracle from the float64-operator oracle -- the noise floor of ANY
    float32 evaluation of this network -- and never looser than 1e-4 of the largest correction."""
    corr, c32, c64 = model.pos_correction.cpu().numpy(), ref.pos_correction, ref64.pos_correction
    floor = _rel(c32, c64)
    err = _rel(corr, c64)
    bar = min(max(3 * floor, 2e-6), 1e-4)
    assert err <= bar, f"{what}: correction rel err {err:.2e} vs the f64 oracle (bar {bar:.1e}; f32 oracle floor {floor:.1e})"
    return err, floor


def test_liquid3d_40cube_bench_density(dev, monkeypatch):
    """Model-level parity where the coarse layers see an INTERIOR: a 40^3 = 64,000-particle box at the bench's density
    (2.0 units wide against R = 0.4: rows with ~33 / ~265 / ~2100 neighbours), Liquid3d with the reference's weights, default
    kernel dispatch, the lattice form of the four lattice -> lattice layers ON and OFF, per-step parity from the oracle's
    states and a free-running 3-step rollout against the free-running oracle."""
    from oracle.model_ref import ModelRef
    from dmcf_amd.pipelines import Simulator
    from dmcf_amd import ops
    from tools import configs, scenes
    w = dict(np.load(os.path.join(GOLDEN, "liquid3d_weights.npz")))
    cfg = configs.LIQUID3D
    scene = scenes.box_scene(40)
    ref, ref64 = ModelRef(cfg, w), ModelRef(cfg, w, f64=True)
    states = [scenes.model_inputs(scene)]
    refs = []
    for s in range(3):
        pos_ref, vel_ref = ref.step(states[-1])
        pos64, vel64 = ref64.step(states[-1])
        refs.append(dict(pos=pos_ref, vel=vel_ref, pos64=pos64, vel64=vel64, c32=ref.pos_correction.copy(),
                         c64=ref64.pos_correction.copy()))
        states.append([pos_ref, vel_ref] + states[-1][2:])
    assert ref.pairs > 2.0e8  # the oracle walked the big lists (s0 <-> s1 / s2 at R = 0.2 / 0.4)

    class R:  # what _check_correction reads
        pass
    for lattice_on in ("1", "0"):
        monkeypatch.setenv("DMCF_LATTICE_CONV", lattice_on)
        model = _build(cfg, w, dev)
        sim = Simulator(model, device="cuda")
        for s in range(3):
            data_t = [None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in states[s]]
            ops.timer = ops.LaunchTimer()
            out = sim.step([data_t])[0]
            recs, ops.timer = ops.timer.results(), None
            n_lat = sum(1 for k, m, _ in recs if k == "cconv" and m.get("lattice"))
            assert n_lat == (4 if lattice_on == "1" else 0), (lattice_on, n_lat)
            pos, vel = out[0].cpu().numpy(), out[1].cpu().numpy()
            r = refs[s]
            assert _rel(pos, r["pos"]) <= 1e-5, f"lattice {lattice_on} step {s}: pos {_rel(pos, r['pos']):.2e}"
            floor = _rel(r["vel"], r["vel64"])
            ulp_floor = 4 * np.finfo(np.float32).eps * np.abs(r["pos64"]).max() / cfg["timestep"] / np.abs(r["vel64"]).max()
            assert _rel(vel, r["vel64"]) <= max(1e-5, 3 * floor, ulp_floor)
            a, b = R(), R()
            a.pos_correction, b.pos_correction = r["c32"], r["c64"]
            err, cfloor = _check_correction(model, a, b, f"lattice {lattice_on} step {s}")
            print(f"40^3 lattice={lattice_on} step {s}: pos {_rel(pos, r['pos']):.2e}, correction {err:.2e} (f32 oracle floor {cfloor:.2e})")
        # free running: the HIP path feeds itself, the oracle fed itself
        data_t = [None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in states[0]]
        for s in range(3):
            data_t = sim.step([data_t])[0]
        drift = _rel(data_t[0].cpu().numpy(), refs[2]["pos"])
        print(f"40^3 lattice={lattice_on}: free-running 3-step drift {drift:.2e}")
        assert drift <= 1e-5


def test_lattice_form_switches_off_far_from_the_origin(dev):
    """The lattice form uses d * voxel where the reference subtracts two rounded positions; the difference grows with
    |x| / extent.  dmcf_amd.lattice.MAX_X_OVER_EXTENT keeps the neighbour-list form (the reference's arithmetic) beyond it:
    the same box at the origin takes the lattice form for its four lattice -> lattice layers, at |x| ~ 50 for none, and
    there the step still matches the oracle."""
    from oracle.model_ref import ModelRef
    from dmcf_amd.pipelines import Simulator
    from dmcf_amd import lattice, ops
    from tools import configs, scenes
    w = dict(np.load(os.path.join(GOLDEN, "liquid3d_weights.npz")))
    cfg = configs.LIQUID3D
    assert 50.0 / 0.8 > lattice.MAX_X_OVER_EXTENT > 1.0 / 0.4
    for origin, expect in (((0.0, 0.0, 0.0), 4), ((50.0, 50.0, 50.0), 0)):
        scene = scenes.box_scene(16, origin=origin, seed=11)
        model = _build(cfg, w, dev)
        sim = Simulator(model, device="cuda")
        data_t = scenes.model_inputs(scene, device=dev)
        ops.timer = ops.LaunchTimer()
        out = sim.step([data_t])[0]
        recs, ops.timer = ops.timer.results(), None
        assert sum(1 for k, m, _ in recs if k == "cconv" and m.get("lattice")) == expect
        ref, ref64 = ModelRef(cfg, w), ModelRef(cfg, w, f64=True)
        data_np = scenes.model_inputs(scene)
        pos_ref, _ = ref.step(data_np)
        ref64.step(data_np)
        assert _rel(out[0].cpu().numpy(), pos_ref) <= 1e-5
        # at |x| ~ 50 one ulp of a position is 4e-6: of the same order as the corrections' own float32 noise; the floor-based
        # bar of _check_correction accounts for it
        _check_correction(model, ref, ref64, f"origin {origin}")


@pytest.mark.parametrize("mode", sorted(FRS_MODES))
def test_column_config1_on_reference_generated_scenes(dev, mode):
    with _frs_mode(mode):
        _column_config1(dev)


def _column_config1(dev):
    """BASELINE.json config 1: configs/column/hrnet.yml (HRNet, kernel [1, 8, 1], 4 scales, 7 fluid features: use_acc
    defaults to True) on the two TEST scenes the REFERENCE's generator produces (tests/golden/column_test.npz, made by
    tests/golden/make_column_fixture.py from datasets/column_gen.py): 200-step rollouts, seeded stand-in weights (no
    checkpoint is shipped for this config).  Every step is checked against the oracle fed with the HIP path's own state, and
    the free-running oracle rollout is compared at the end."""
    from oracle.model_ref import ModelRef
    from dmcf_amd.pipelines import Simulator
    from tools import configs, scenes
    cfg = dict(configs.COLUMN_HRNET)
    w = scenes.random_weights(cfg, seed=2)
    fix = np.load(os.path.join(GOLDEN, "column_test.npz"))
    for s in (0, 1):
        pos0, vel0, grav0 = fix[f"s{s}_pos"][0], fix[f"s{s}_vel"][0], fix[f"s{s}_grav"][0]
        box, nrm = fix[f"s{s}_box"], fix[f"s{s}_box_normals"]
        acc = np.broadcast_to(grav0, pos0.shape).astype(np.float32).copy()  # get_rollout: grav per frame -> per particle
        model = _build(cfg, w, dev)
        sim = Simulator(model, device="cuda")
        ref = ModelRef(cfg, w)
        free = ModelRef(cfg, w)
        state_np = [pos0, vel0, acc, None, box, nrm]
        free_np = list(state_np)
        worst = 0.0
        for t in range(200):
            data_t = [None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in state_np]
            out = sim.step([data_t])[0]
            pos_ref, vel_ref = ref.step(state_np)
            pos = out[0].cpu().numpy()
            assert np.isfinite(pos).all()
            scale = max(np.abs(pos_ref).max(), 1e-30)
            worst = max(worst, np.abs(pos - pos_ref).max() / scale)
            assert np.abs(pos - pos_ref).max() <= 1e-5 * scale, f"scene {s} step {t}"
            if t in (0, 100, 199):
                _displacement_check(state_np[0], pos, pos_ref, f"column scene {s} step {t}")
            state_np = [pos, out[1].cpu().numpy()] + state_np[2:]
            p, v = free.step(free_np)
            free_np = [p, v] + free_np[2:]
        drift = np.abs(state_np[0] - free_np[0]).max() / np.abs(free_np[0]).max()
        print(f"column scene {s}: worst per-step rel err {worst:.2e}, free-running drift after 200 steps {drift:.2e}")
        assert drift <= 1e-4


@pytest.mark.parametrize("mode", sorted(FRS_MODES))
def test_liquid3d_real_weights_box_scene(dev, mode):
    from tools import configs, scenes
    w = dict(np.load(os.path.join(GOLDEN, "liquid3d_weights.npz")))
    scene = scenes.box_scene(12)
    model, ref = _compare_step(configs.LIQUID3D, w, scene, dev, steps=3, mode=mode)
    # the searches the reference would run per step: 18; distinct ones actually run: 12
    assert len(model._all_convs) == 18


def test_fused_input_convs_equal_the_two_layers(dev, monkeypatch):
    """Inside a step the two input layers run as one block-diagonal convolution on the all -> all list
    (PBFNet._fused_input_convs): same features, same fluid-neighbour counts, same step as with DMCF_FUSE_INPUT_CONVS=0."""
    from tools import configs, scenes
    from dmcf_amd.utils.convolutions import neighbor_cache
    w = dict(np.load(os.path.join(GOLDEN, "liquid3d_weights.npz")))
    res = {}
    for fuse in ("1", "0", "1 no boundary", "0 no boundary"):
        monkeypatch.setenv("DMCF_FUSE_INPUT_CONVS", fuse[0])
        model = _build(configs.LIQUID3D, w, dev)
        scene = scenes.box_scene(14, seed=5)
        if "no boundary" in fuse:  # every boundary particle is cropped away: the fused form must cope with an empty set
            scene["box"] = scene["box"] + np.float32(100.0)
        data = scenes.model_inputs(scene, device=dev)
        with neighbor_cache():
            d = model.transform(data)
            x = model.preprocess(d)
            assert (model.fluid_convs.nns is None) == (fuse[0] == "1")
            out = model.run_forward(x, d)
            pos, vel = model.postprocess(out, d, training=False)
        res[fuse] = (x[1].cpu().numpy(), model.num_fluid_neighbors.cpu().numpy(), pos.cpu().numpy())
    for tag in ("", " no boundary"):
        feats, counts, pos = res["1" + tag]
        feats0, counts0, pos0 = res["0" + tag]
        assert feats.shape == feats0.shape and (tag == "" or feats.shape[0] == 14 ** 3)
        assert np.abs(feats - feats0).max() <= 2e-6 * np.abs(feats0).max()
        np.testing.assert_array_equal(counts, counts0)
        assert counts.min() >= 1 and counts.max() > 20  # every particle is its own neighbour; the bulk has ~30
        assert _rel(pos, pos0) <= 1e-6


def test_cross_layer_paired_convs_equal_the_two_layers(dev, monkeypatch):
    """HRNet._paired_convs: conv200_2 (layer 2, s2 -> s0) and conv300_2 (layer 3, s2 -> s0) of the Liquid3d net share one
    neighbour list and run as ONE block-diagonal launch from the second step on -- same step as with the two launches."""
    from dmcf_amd import ops
    from dmcf_amd.pipelines import Simulator
    from tools import configs, scenes
    w = dict(np.load(os.path.join(GOLDEN, "liquid3d_weights.npz")))
    res = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("DMCF_FUSE_CROSS_LAYER", fuse)
        sim = Simulator(_build(configs.LIQUID3D, w, dev), device="cuda")
        state = scenes.model_inputs(scenes.box_scene(20, seed=9), device=dev)
        launches = []
        for _ in range(3):
            ops.timer = ops.LaunchTimer()
            state = sim.step([state])[0]
            recs, ops.timer = ops.timer.results(), None
            launches.append(sum(1 for k, m, _ in recs if k == "cconv"))
        res[fuse] = (state[0].cpu().numpy(), state[1].cpu().numpy(), launches)
    # 18 layers, the input pair fused; with the cross-layer pairs conv200_2 + conv300_2 (s2 -> s0, one pass of the class-sorted
    # kernel) and conv200_1 + conv300_1 (s1 -> s0, 8 + 16 channels: one walk of the pair kernel) two launches less
    assert res["0"][2] == [17, 17, 17] and res["1"][2] == [15, 15, 15], (res["0"][2], res["1"][2])
    assert _rel(res["1"][0], res["0"][0]) <= 1e-6, _rel(res["1"][0], res["0"][0])
    assert _rel(res["1"][1], res["0"][1]) <= 1e-4, _rel(res["1"][1], res["0"][1])


def test_layer_sums_in_the_conv_epilogue_equal_the_separate_terms(dev, monkeypatch):
    """HRNet forms a layer's sum conv_0 + conv_1 + conv_2 + dense + residual in ONE buffer (Dense.product with the residual as
    the GEMM's C operand, every convolution adding in its epilogue, DMCF_FLAG_ACCUMULATE) instead of a tensor per term and an
    elementwise kernel per '+': the same terms in another order -- three steps agree to rounding, with fewer launches outside
    the library."""
    from dmcf_amd.models import hrnet
    from dmcf_amd.pipelines import Simulator
    from tools import configs, scenes
    w = dict(np.load(os.path.join(GOLDEN, "liquid3d_weights.npz")))
    res = {}
    for fuse in (True, False):
        monkeypatch.setattr(hrnet, "_FUSE_EPILOGUE", fuse)
        sim = Simulator(_build(configs.LIQUID3D, w, dev), device="cuda")
        state = scenes.model_inputs(scenes.box_scene(20, seed=9), device=dev)
        for _ in range(3):
            state = sim.step([state])[0]
        convs = [c for _, c in sim.model._all_convs]
        assert all(c.accumulate_into is None and c.extra_bias is None for c in convs)  # one-shot requests, all taken
        res[fuse] = (state[0].cpu().numpy(), state[1].cpu().numpy())
    assert _rel(res[True][0], res[False][0]) <= 1e-6, _rel(res[True][0], res[False][0])
    assert _rel(res[True][1], res[False][1]) <= 1e-4, _rel(res[True][1], res[False][1])


def test_liquid3d_momentum_conservation(dev):
    """ASCC head: sum of the network output over fluid + boundary particles vanishes (SURVEY section 4 invariant 4)."""
    from tools import configs, scenes
    from dmcf_amd.utils.convolutions import neighbor_cache
    w = dict(np.load(os.path.join(GOLDEN, "liquid3d_weights.npz")))
    model = _build(configs.LIQUID3D, w, dev)
    data = scenes.model_inputs(scenes.box_scene(16, seed=3), device=dev)
    with neighbor_cache():
        d = model.transform(data)
        x = model.preprocess(d)
        out = model.run_forward(x, d)
    tot = out.double().sum(0).abs()
    assert torch.all(tot <= 2e-5 * out.double().abs().sum(0)), tot


def test_waterramps_arch_random_weights_2d(dev):
    from tools import configs, scenes
    w = scenes.random_weights(configs.WATERRAMPS, seed=0)
    scene = scenes.box_scene(36, h=0.005, dim=2, origin=(-0.09, -0.09, 0.0))
    _compare_step(configs.WATERRAMPS, w, scene, dev, steps=2)


def test_wbcsph_arch_random_weights_grav_eqvar(dev):
    from tools import configs, scenes
    w = scenes.random_weights(configs.WBC_SPH, seed=1)
    scene = scenes.box_scene(40, h=0.0025, dim=2)
    g = np.float32([3.0, -9.0, 0.0])  # tilted gravity exercises align_vector / the inverse transform
    _compare_step(configs.WBC_SPH, w, scene, dev, grav=g, steps=2)


def test_column_hrnet_1d(dev):
    from tools import configs, scenes
    w = scenes.random_weights(configs.COLUMN_HRNET, seed=2, fluid_channels=4)
    cfg = dict(configs.COLUMN_HRNET, use_acc=False)
    y = (np.arange(40, dtype=np.float32) + 0.5) * np.float32(0.005)
    pos = np.stack([np.zeros_like(y), y, np.zeros_like(y)], -1)
    scene = dict(pos=pos, vel=np.zeros_like(pos), box=np.float32([[0, -0.0025, 0], [0, -0.0075, 0]]),
                 box_normals=np.float32([[0, 1, 0], [0, 1, 0]]))
    _compare_step(cfg, w, scene, dev, steps=2)


def test_cconv_baseline_model_2d(dev):
    from tools import configs, scenes
    cfg = dict(configs.CCONV2D, use_acc=False)
    w = scenes.random_weights(cfg, seed=3, fluid_channels=4)
    scene = scenes.box_scene(30, h=0.0125, dim=2)
    _compare_step(cfg, w, scene, dev, steps=2)


def test_rollout_with_changing_particle_count(dev):
    """run_sample.py:173-177 appends inflow particles between steps: nothing may assume a fixed N."""
    from dmcf_amd.pipelines import Simulator
    from tools import configs, scenes
    w = dict(np.load(os.path.join(GOLDEN, "liquid3d_weights.npz")))
    model = _build(configs.LIQUID3D, w, dev)
    sim = Simulator(model, device="cuda")
    scene = scenes.box_scene(8)
    res = sim.run_rollout([dict(pos=scene["pos"][None], vel=scene["vel"][None], grav=[None], box=scene["box"][None],
                                box_normals=scene["box_normals"][None])], timesteps=3)
    assert len(res[0]) == 3 and res[0][2][0].shape == (512, 3)
    state = res[0][-1]
    extra = torch.from_numpy(scenes.box_scene(4, origin=(0.1, 0.1, 0.1), seed=5)["pos"]).to(dev)
    state = [torch.cat([state[0], extra]), torch.cat([state[1], torch.zeros_like(extra)])] + list(state[2:])
    out = sim.step([state])[0]
    assert out[0].shape == (512 + 64, 3) and torch.isfinite(out[0]).all()


def test_estimated_neighbour_buffers_recover_from_overflow(dev):
    """Steps after the first enqueue their searches with buffer sizes estimated from the previous step (no host
    round trip).  Feeding a much larger scene next must be detected and the step repeated exactly."""
    from dmcf_amd.pipelines import Simulator
    from dmcf_amd.utils.convolutions import neighbor_hints
    from tools import configs, scenes
    w = dict(np.load(os.path.join(GOLDEN, "liquid3d_weights.npz")))
    model = _build(configs.LIQUID3D, w, dev)
    sim = Simulator(model, device="cuda")
    small = scenes.model_inputs(scenes.box_scene(6, seed=1), device=dev)
    big = scenes.model_inputs(scenes.box_scene(12, seed=2), device=dev)
    sim.step([small])
    sim.step([small])  # now running on estimates
    assert any(h is not None for h in neighbor_hints())
    out = sim.step([big])[0]  # every list overflows its estimate -> repeated with exact sizes
    fresh = Simulator(_build(configs.LIQUID3D, w, dev), device="cuda")
    neighbor_hints().clear()
    ref = fresh.step([big])[0]
    assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
    # and the estimate path itself is exact when nothing overflows
    a = sim.step([big])[0]
    neighbor_hints().clear()
    b = fresh.step([big])[0]
    assert torch.equal(a[0], b[0])


def test_lists_of_a_scene_with_spray_take_the_estimated_csr_form(dev):
    """A bulk whose rows hold thousands of entries next to thousands of isolated droplets (the 100k dam break after ~40 steps):
    padded rows -- every query reserves the longest row's stride -- would be mostly air (34 GB per step there).  From the
    second step on the per-step cache sees that from the previous step's pair counts and enqueues count + scan + write into
    buffers sized from them instead (no host round trip either): same positions bit for bit as the exact searches, no repeated
    step, and a fraction of the memory."""
    from dmcf_amd import ops
    from dmcf_amd.pipelines import Simulator
    from dmcf_amd.utils.convolutions import neighbor_hints
    from tools import configs, scenes
    w = dict(np.load(os.path.join(GOLDEN, "liquid3d_weights.npz")))
    scene = scenes.box_scene(14, seed=3)
    rng = np.random.default_rng(8)
    g = np.stack(np.meshgrid(*[np.arange(16)] * 3, indexing="ij"), -1).reshape(-1, 3)[:3000]
    spray = (np.float32([3.0, 3.0, 3.0]) + np.float32(1.1) * g + rng.uniform(-0.05, 0.05, size=(3000, 3))).astype(np.float32)
    scene = dict(scene, pos=np.concatenate([scene["pos"], spray]), vel=np.concatenate([scene["vel"], np.zeros_like(spray)]))
    sim = Simulator(_build(configs.LIQUID3D, w, dev), device="cuda")
    state = scenes.model_inputs(scene, device=dev)
    outs, kinds = [], []
    for _ in range(3):
        torch.cuda.reset_peak_memory_stats(dev)
        ops.timer = ops.LaunchTimer()
        state = sim.step([state])[0]
        recs, ops.timer = ops.timer.results(), None
        kinds.append([k for k, _, _ in recs if k.startswith("frs_")])
        outs.append((state[0].clone(), torch.cuda.max_memory_allocated(dev)))
    assert sim.repeated_steps == 0
    # steps 2 and 3 run on estimates (every class of step 3 was searched in step 2: nothing falls back to the exact form): some
    # lists padded, some as count + write into estimated buffers
    assert "frs_search_padded" in kinds[2] and "frs_write" in kinds[2] and kinds[1] == kinds[2], kinds
    fresh = Simulator(_build(configs.LIQUID3D, w, dev), device="cuda")
    neighbor_hints().clear()
    ref = scenes.model_inputs(scene, device=dev)
    os.environ["DMCF_NO_ESTIMATE"] = "1"
    try:
        for t in range(3):
            ref = fresh.step([ref])[0]
            assert torch.equal(ref[0], outs[t][0]), f"step {t}"
    finally:
        os.environ.pop("DMCF_NO_ESTIMATE")
    print("peak allocated per step (MiB):", [round(m / 2 ** 20) for _, m in outs])


def test_density_feature_flags_2d(dev):
    """dens_feats + pres_feats + dens_norm (pbf_model.py:351-365,421-431; hrnet.py:87-89) on the WaterRamps
    architecture: fused density kernel, PointSampling to the coarse scales, doubled layer inputs."""
    from tools import configs, scenes
    cfg = dict(configs.WATERRAMPS, dens_feats=True, pres_feats=True, dens_norm=True, window_dens="poly6", rest_dens=12.0)
    w = scenes.random_weights(cfg, seed=4)
    scene = scenes.box_scene(36, h=0.005, dim=2, origin=(-0.09, -0.09, 0.0))
    model, ref = _compare_step(cfg, w, scene, dev, steps=2)
    assert ref.dens is not None


def test_equivar_scaled_mean_offset_head_2d(dev):
    """``equivar: True`` (pbf_model.py:183-189,456-463; losses.py:337-364; no shipped config sets it): the network's output goes
    through Dense(1) and scales the mean offset to the neighbours within the first radius (the quaternion branch is commented
    out in the reference: rot = None)."""
    from tools import configs, scenes
    cfg = dict(configs.WATERRAMPS, equivar=True)
    w = scenes.random_weights(cfg, seed=11)
    cout = w["model/sym_convs/0/kernel"].shape[-1]
    rng = np.random.default_rng(12)
    w["model/scale_dens/kernel"] = rng.uniform(-0.5, 0.5, size=(cout, 1)).astype(np.float32)
    w["model/scale_dens/bias"] = np.float32([0.7])
    scene = scenes.box_scene(30, h=0.005, dim=2, origin=(-0.07, -0.07, 0.0))
    model, ref = _compare_step(cfg, w, scene, dev, steps=2)
    assert model.equivar and float(model.pos_correction.abs().max()) > 0


def test_canyon_sample_with_inflow(dev):
    """The reference's demo (run_sample.py: canyon scene, Liquid3d checkpoint, inflow every other step) on frames of the
    reference's own scene file (tests/golden/canyon_crop.msgpack.zst): every step of dmcf_amd.run_sample.run_rollout
    against the oracle model fed with the same state."""
    from oracle.model_ref import ModelRef
    from dmcf_amd.datasets import read_scene
    from dmcf_amd.pipelines import Simulator
    from dmcf_amd.run_sample import INFLOW_VELOCITY, run_rollout
    from tools import configs
    w = dict(np.load(os.path.join(GOLDEN, "liquid3d_weights.npz")))
    frame0 = read_scene(os.path.join(GOLDEN, "canyon_crop.msgpack.zst"))[0]
    model = _build(configs.LIQUID3D, w, dev)
    sim = Simulator(model, device="cuda")
    results, timing = run_rollout(sim, frame0, timesteps=6, inflow=4)
    assert [r.shape[0] for r in results] == [1280, 1280, 1280, 2560, 2560, 3840] and len(timing) == 5
    ref = ModelRef(configs.LIQUID3D, w)
    in_pos = frame0["pos"]
    in_vel = (frame0["vel"] + np.float32([INFLOW_VELOCITY])).astype(np.float32)
    in_acc = np.zeros_like(in_pos) + np.float32([[0, configs.LIQUID3D.get("grav", -9.81), 0]])
    data = [in_pos, in_vel, in_acc, None, frame0["box"], frame0["box_normals"]]
    for t in range(5):
        # oracle step from the HIP path's own previous state: per-step parity, no error accumulation
        pos_ref, vel_ref = ref.step(data)
        got = results[t + 1].cpu().numpy()
        assert got.shape == pos_ref.shape
        assert _rel(got, pos_ref) <= 1e-5, f"step {t}: {_rel(got, pos_ref):.2e}"
        data = [got, vel_ref] + data[2:]
        if 4 > t and t % 2 == 1:
            data = [np.concatenate([data[0], in_pos]), np.concatenate([data[1], in_vel]), np.concatenate([data[2], in_acc]),
                    None, data[4], data[5]]


def test_run_pipeline_test_split_on_canyon_frames(dev, tmp_path):
    """run_pipeline.py --split test (run_pipeline.py:80-154 -> Simulator.run_test, simulator.py:111-165): YAML -> model ->
    checkpoint -> get_rollout -> run_rollout -> write_results, on the frames of the reference's canyon scene."""
    import yaml
    from dmcf_amd import run_pipeline
    from dmcf_amd.datasets import read_scene
    from dmcf_amd.pipelines import Simulator
    from tools import configs
    w = dict(np.load(os.path.join(GOLDEN, "liquid3d_weights.npz")))
    cfg = dict(dataset=dict(name="CConvData3D"),
               model=dict(configs.LIQUID3D, ckpt_path=None),
               pipeline=dict(name="Simulator", version="v0", main_log_dir=str(tmp_path / "logs"), output_dir=str(tmp_path / "out"),
                             data_generator=dict(scale=[1.0, 1.0, 1.0], train=dict(stride=1), valid=dict(stride=1),
                                                 test=dict(stride=1, time_start=0, time_end=50))))
    yml = tmp_path / "liquid3d.yml"
    yml.write_text(yaml.safe_dump(cfg))
    args, extra = run_pipeline.parse_args(["-c", str(yml), "--split", "test", "--dataset_path", GOLDEN, "--pipeline.version", "v7"])
    pipe = run_pipeline.build(args, extra)
    assert isinstance(pipe, Simulator) and pipe.cfg.out_dir.endswith("SymNet_CConvData3D_v7")
    from dmcf_amd.utils import tf_checkpoint as tc
    tc.load_into_model(pipe.model, w, device=dev)  # (the checkpoint blob itself is not shipped to the GPU box)
    paths = pipe.run_test(epoch=3)
    assert len(paths) == 1 and os.path.basename(paths[0]).startswith("0003.")
    frames = read_scene(os.path.join(GOLDEN, "canyon_crop.msgpack.zst"))
    assert paths[0].endswith("0003.hdf5")  # an HDF5 file, as the reference writes (simulator.py:155)
    # read back by the HDF5 C library where the image has it, else by the structural walk (tests/test_hdf5_writer.py)
    import test_hdf5_writer as h5t
    L = h5t._libhdf5()
    if L is not None:
        got = h5t._read_with_libhdf5(L, paths[0])
        assert list(got) == ["SymNet"] and got["SymNet"]["pred"][1]["type"] == "PARTICLE"
        pred, gt, bnd = (got["SymNet"][k][0] for k in ("pred", "gt", "bnd"))
    else:
        got = h5t._walk(paths[0])
        pred, gt, bnd = (got["SymNet"][k] for k in ("pred", "gt", "bnd"))
    assert pred.shape == (3, frames[0]["pos"].shape[0], 3) and gt.shape == pred.shape and bnd.shape == frames[0]["box"].shape
    np.testing.assert_array_equal(pred[0], frames[0]["pos"])
    np.testing.assert_array_equal(gt[2], frames[2]["pos"])
    # the rollout is the Simulator's: step 1 equals one run_inference on frame 0
    sim = Simulator(_build(configs.LIQUID3D, w, dev), device="cuda")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)  # noqa: E731
    grav = t(np.broadcast_to(frames[0]["grav"], frames[0]["pos"].shape)) if frames[0].get("grav") is not None else None
    one = sim.step([[t(frames[0]["pos"]), t(frames[0]["vel"]), grav, None, t(frames[0]["box"]), t(frames[0]["box_normals"])]])[0]
    np.testing.assert_array_equal(pred[1], one[0].cpu().numpy())
    assert np.isfinite(pred).all()


def test_against_reference_rollout_golden(dev):
    """The HIP path against a rollout captured from the REFERENCE itself (tools/capture_golden.py --reference, run off-box on
    TensorFlow 2.5 + Open3D 0.15.2): 10 free-running steps of the Liquid3d SymNet on the canyon scene, positions within 1e-5
    relative per step.  Skipped until the capture exists -- until then parity is "vs our restatement" (DESIGN.md section 2)."""
    path = os.path.join(GOLDEN, "open3d_golden.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/open3d_golden.npz not captured yet (parity unpinned)")
    g = np.load(path)
    if "rollout_pos" not in g:
        pytest.skip("open3d_golden.npz was captured without --reference")
    from dmcf_amd.pipelines import Simulator
    from tools import configs
    w = dict(np.load(os.path.join(GOLDEN, "liquid3d_weights.npz")))
    sim = Simulator(_build(configs.LIQUID3D, w, dev), device="cuda")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)  # noqa: E731
    state = [t(g["rollout_pos"][0]), t(g["rollout_vel0"]), None, None, t(g["rollout_box"]), t(g["rollout_box_normals"])]
    for k in range(1, g["rollout_pos"].shape[0]):
        state = sim.step([state])[0]
        ref = g["rollout_pos"][k]
        assert _rel(state[0].cpu().numpy(), ref) <= 1e-5 * k, f"free-running step {k}"


def test_fps_multiscale_2d(dev):
    """voxel_size: None: farthest-point-sampled scales (losses.py:274-282) and HRNet's cross-scale Dense branch
    (hrnet.py:100-113) on the WaterRamps architecture."""
    from tools import configs, scenes
    cfg = dict(configs.WATERRAMPS, voxel_size=None, centralize=False)
    w = scenes.random_weights(cfg, seed=6)
    scene = scenes.box_scene(30, h=0.005, dim=2, origin=(-0.07, -0.07, 0.0))
    model, ref = _compare_step(cfg, w, scene, dev, steps=2)
    assert [len(i) for i in ref.fps_idx[1:]] == [len(model.all_pos) // 2, len(model.all_pos) // 4]


def test_bench_line_contract():
    """bench.py prints ONE JSON line with the fields the driver reads (small workload so the test stays quick)."""
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--side", "24", "--steps", "2", "--warmup", "1",
                          "--cpu-side", "8"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "roofline_groups", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["vs_baseline"] is None and d["value"] > 0
    assert abs(d["value"] - 24 ** 3 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and d["roofline"]["bound"] == "hbm"
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and d["cpu_baseline"]["kind"] == "port"
    assert "workload" in d["config"] and "model" not in d["config"]
    g = d["roofline_groups"]
    # 18 layers per step: 4 in the lattice form, the two input layers as one launch, conv200_2 + conv300_2 and conv200_1 + conv300_1
    # as one launch each
    assert g["neighbour_list"]["launches"] == 2 * 11 and g["lattice"]["launches"] == 2 * 4
    assert d["roofline"]["kernel"].startswith("dmcf::cconv_") and d["roofline"]["kernel"][6:] in g["by_kernel"]
    assert 0 < g["lattice"]["frac"] < 1 and 0 < g["neighbour_list"]["frac"] < 1
    # the fractions that bound (round 5): the dominant kernel and every neighbour-list kernel carry their flops against the f32
    # matrix peak; the DRAM fraction is there when profiles/ holds counter traffic for the kernel
    assert 0 < d["roofline"]["frac_flops"] < 1 and "frac_dram" in d["roofline"]
    assert all(0 < v["frac_flops"] < 1 for v in g["by_kernel"].values() if "frac_mfma_f32" not in v)
    assert 0 < g["neighbour_list"]["frac_flops"] < 1
    assert "frs_query" not in d["kernel_ms_per_step"]


def test_full_size_step_against_the_oracle(dev):
    """BASELINE.json's bench configuration at FULL size against the CPU oracle: one step of the 100^3-particle box
    (1,124,864 points, 3.4e9 neighbour pairs; about a minute of the host's cores).  Positions within 1e-5 as north_star
    asks, and the network output within the float32 noise the 40^3 test measures (a few 1e-6 of the largest correction)."""
    from oracle.model_ref import ModelRef
    from dmcf_amd.pipelines import Simulator
    from tools import configs, scenes
    w = dict(np.load(os.path.join(GOLDEN, "liquid3d_weights.npz")))
    scene = scenes.box_scene(100)
    model = _build(configs.LIQUID3D, w, dev)
    sim = Simulator(model, device="cuda")
    out = sim.step([scenes.model_inputs(scene, device=dev)])[0]
    ref = ModelRef(configs.LIQUID3D, w)
    pos_ref, vel_ref = ref.step(scenes.model_inputs(scene))
    assert ref.pairs > 3.0e9
    assert _rel(out[0].cpu().numpy(), pos_ref) <= 1e-5
    _displacement_check(scene["pos"], out[0].cpu().numpy(), pos_ref, "full size (config 5, one GPU)")
    cerr = _rel(model.pos_correction.cpu().numpy(), ref.pos_correction)
    print(f"full size: pos {_rel(out[0].cpu().numpy(), pos_ref):.2e}, correction {cerr:.2e} (vs the float32 oracle)")
    assert cerr <= 2e-5


def test_full_size_step_on_the_degraded_bench_scene_against_the_oracle(dev):
    """What bench.py actually times: the driver's window is steps 6 .. 25 of the 1M-particle rollout, by whose end ~15 % of the
    fluid has leaked through the 2-layer shell at up to ~85 m/s -- stray particles far from the box,
    grid_pos in its sort-based form, the lattice layers back on neighbour lists, rows that outgrew their strides.  25 steps on
    the GPU, then ONE step from that state against the CPU oracle fed with the same state (about a minute of the host's cores);
    the step runs on the rollout's own Simulator, i.e. on buffers sized from the previous steps."""
    from oracle.model_ref import ModelRef
    from dmcf_amd.pipelines import Simulator
    from tools import configs, scenes
    w = dict(np.load(os.path.join(GOLDEN, "liquid3d_weights.npz")))
    scene = scenes.box_scene(100)
    model = _build(configs.LIQUID3D, w, dev)
    sim = Simulator(model, device="cuda")
    state = scenes.model_inputs(scene, device=dev)
    for _ in range(25):
        state = sim.step([state])[0]
    lo, hi = torch.tensor(scene["box"].min(axis=0), device=dev), torch.tensor(scene["box"].max(axis=0), device=dev)
    outside = int(((state[0] < lo) | (state[0] > hi)).any(dim=1).sum())
    speed = float(state[1].norm(dim=1).max())
    assert outside > 10000 and speed > 10.0, (outside, speed)  # the scene HAS degraded (if this ever stops, the test has lost its point)
    before = [None if x is None else x.cpu().numpy() for x in state]
    import oracle
    ref = ModelRef(configs.LIQUID3D, w)
    # This state holds the case that separates the neighbour sets (ops.SEARCH_SETS): a fluid particle a rounding step from the
    # middle of a hash voxel (z = 1.3, R = 0.1), whose row open3d's float walk truncates.  The product's default (the set of
    # the distance test) against the oracle walking all 27 voxels; the emulation of the walk against the oracle's default.
    for name, bins in sorted(FRS_MODES.items()):
        with _frs_mode(name):
            out = sim.step([state])[0]
            pos_ref, vel_ref = ref.step(before)
        perr = _rel(out[0].cpu().numpy(), pos_ref)
        cerr = _rel(model.pos_correction.cpu().numpy(), ref.pos_correction)
        print(f"degraded bench scene, step 26 [{name} vs bins={bins}]: {outside} particles outside the shell, max speed {speed:.1f} m/s, "
              f"{ref.pairs:.3g} pairs; pos {perr:.2e}, correction {cerr:.2e} (vs the float32 oracle)")
        assert torch.isfinite(out[0]).all() and perr <= 1e-5
        # (corrections reach centimetres here; their float32 noise is held to 5e-5 of the largest below -- 1e-5 of the largest, as an
        # absolute number, enters the per-particle bar)
        _displacement_check(before[0], out[0].cpu().numpy(), pos_ref, f"degraded bench scene [{name}]",
                            noise=1e-5 * float(np.abs(ref.pos_correction).max()))
        assert cerr <= 5e-5


@pytest.mark.parametrize("mode", sorted(FRS_MODES))
@pytest.mark.parametrize("name,steps", [("liquid3d_dam", 60), ("waterramps", 60), ("wbcsph", 60)])
def test_shortened_rollouts_of_configs_2_3_4(dev, name, steps, mode):
    with _frs_mode(mode):
        _shortened_rollout(dev, name, steps, momentum=mode == "distance")


def _shortened_rollout(dev, name, steps, momentum):
    """BASELINE.json configs 2 / 3 / 4 (README.md:79: 600 / 3200 / 200 frames; tools/long_rollout.py runs them at full length,
    profiles/r0N_long_rollouts.md) as 60-step rollouts inside the suite: every step finite, steps 0 / 30 / 59 against the CPU
    oracle fed with the HIP path's own state, and the ASCC head's momentum residual at rounding level in EVERY step -- the
    default search returns symmetric lists (the set of the distance test; the emulations of open3d's float walk, under which
    about one query in 10^6 loses part of its row and that particle's pair terms no longer cancel, are opt-in:
    include/dmcf_hip.h).  Both neighbour sets run (FRS_MODES), each against the oracle's statement of it; the momentum bar holds
    under the default only -- under the emulation the lists lose their symmetry exactly where the library's do."""
    from oracle.model_ref import ModelRef
    from dmcf_amd.pipelines import Simulator
    from tools import long_rollout, scenes
    cfg, w, scene, grav = long_rollout.setup(name)
    model = _build(cfg, w, dev)
    sim = Simulator(model, device="cuda")
    ref, ref64 = ModelRef(cfg, w), ModelRef(cfg, w, f64=True)
    state = scenes.model_inputs(scene, device=dev, grav=grav)
    worst_mom = 0.0
    for t in range(steps):
        before = [None if x is None else x.cpu().numpy() for x in state] if t in (0, steps // 2, steps - 1) else None
        state = sim.step([state])[0]
        assert torch.isfinite(state[0]).all() and torch.isfinite(state[1]).all(), f"step {t}"
        out = torch.cat([model.pos_correction, model.obs], dim=0).double()
        mom = float((out.sum(0).abs() / out.abs().sum(0).clamp(min=1e-300)).max())
        worst_mom = max(worst_mom, mom)
        assert not momentum or mom <= 2e-6, f"step {t}: momentum residual {mom:.2e}"
        if before is not None:
            pos_ref, _ = ref.step(before)
            err = _rel(state[0].cpu().numpy(), pos_ref)
            assert err <= 1e-5, f"step {t}: pos rel err {err:.2e}"
            # (the float32 noise floor of the network output, measured: the same restatement with float64 operators)
            ref64.step(before)
            _displacement_check(before[0], state[0].cpu().numpy(), pos_ref, f"{name} step {t}",
                                noise=3 * float(np.abs(ref.pos_correction - ref64.pos_correction).max()))
    print(f"{name}: {steps} steps, worst momentum residual {worst_mom:.2e}, {sim.repeated_steps} repeated")


def test_full_size_step_properties(dev):
    """BASELINE.json's bench configuration at full size (100^3 fluid + 124,864 boundary particles, Liquid3d weights),
    checked through size-independent properties: the step is bit-reproducible (every kernel has a fixed summation
    order), the ASCC head conserves momentum, the estimated-buffer path equals the exact one, nothing is NaN."""
    from tools import configs, scenes
    from dmcf_amd.pipelines import Simulator
    from dmcf_amd.utils.convolutions import neighbor_cache
    w = dict(np.load(os.path.join(GOLDEN, "liquid3d_weights.npz")))
    model = _build(configs.LIQUID3D, w, dev)
    sim = Simulator(model, device="cuda")
    state = scenes.model_inputs(scenes.box_scene(100), device=dev)
    a = sim.step([state])[0]          # exact sizes (first step), fills the estimates
    b = sim.step([state])[0]          # estimated sizes, same input
    c = sim.step([state])[0]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(b[0], c[0])
    assert torch.isfinite(a[0]).all() and torch.isfinite(a[1]).all()
    with neighbor_cache():
        d = model.transform(state)
        x = model.preprocess(d)
        out = model.run_forward(x, d)
    tot = out.double().sum(0).abs()
    assert torch.all(tot <= 2e-5 * out.double().abs().sum(0)), tot
    corr = model.pos_correction
    assert corr.shape == (100 ** 3, 3) and float(corr.abs().max()) < 0.05  # a sub-particle-spacing correction


def test_ascc_head_shares_the_trunk_list(dev, monkeypatch):
    """Liquid3d: from the second step on the ASCC head takes the s0 -> s0 list of the trunk (searched WITH the query points)
    and skips the pairs (i, i) in the kernel (DMCF_FLAG_SKIP_SELF): one search less per step, the same positions bit for
    bit (the other pairs keep their order; a zero-weight pair adds an exact zero)."""
    from dmcf_amd import ops
    from dmcf_amd.pipelines import Simulator
    from tools import configs, scenes
    w = dict(np.load(os.path.join(GOLDEN, "liquid3d_weights.npz")))
    res = {}
    for share in ("1", "0"):
        monkeypatch.setenv("DMCF_SHARE_LISTS", share)
        sim = Simulator(_build(configs.LIQUID3D, w, dev), device="cuda")
        state = scenes.model_inputs(scenes.box_scene(20, seed=9), device=dev)
        searches = []
        for _ in range(3):
            ops.timer = ops.LaunchTimer()
            state = sim.step([state])[0]
            recs, ops.timer = ops.timer.results(), None
            searches.append(sum(1 for k, m, _ in recs if k.startswith("frs_search")))
        res[share] = (searches, state[0].clone())
    assert res["0"][0][1] == res["0"][0][2] and res["1"][0][0] == res["0"][0][0]  # the first step learns the head's kernel
    assert res["1"][0][1] == res["0"][0][1] - 1 and res["1"][0][2] == res["0"][0][2] - 1
    assert torch.equal(res["1"][1], res["0"][1])
