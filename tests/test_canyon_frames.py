"""One model step against the reference's own simulation data.

tests/golden/canyon_frames.npz = frames 7 ... 12 of /root/reference/datasets/canyon_data/canyon.msgpack.zst, the scene
run_sample.py:160-179 steps with checkpoints/Liquid3d (tests/golden/liquid3d_weights.npz).  Frame t + 1 is what those
weights were trained to reach from frame t (models/pbf_model.py:234-250: integrate, then add the network's correction),
so it is ground truth that does not pass through the oracle: a step that lands closer to it than the bare integration,
with a correction pointing the right way, cannot have the filter layout, orientation, interpolation, align_corners,
window argument or ASCC mirror wrong (profiles/r06_reading_sweep.md: each of those readings fails these bars by 1.4 - 20x).
What the frames cannot tell apart (radial vs volume-preserving map, grid_pos constants, the search's voxel set) stays
with tools/capture_golden.py.

Bars (measured on the oracle: ratio 0.64 - 0.75, cosine 0.79 - 0.81): ratio <= 0.8, cosine >= 0.7.
"""
import numpy as np
import pytest

from tools import canyon, configs


@pytest.fixture(scope="module")
def fx():
    return canyon.load()


@pytest.fixture(scope="module")
def weights():
    return dict(np.load(canyon.WEIGHTS))


def test_fixture_is_consecutive_frames(fx):
    assert fx["pos"].shape == (6, 1280, 3) and fx["vel"].shape == (6, 1280, 3) and fx["pos"].dtype == np.float32
    assert list(fx["frame_id"]) == [7, 8, 9, 10, 11, 12]
    assert fx["box"].shape == fx["box_normals"].shape and fx["box"].shape[0] > 10000
    # the boundary crop contains the model's own crop (fluid bounding box +- the largest filter extent, pbf_model.py:330-336)
    lo, hi = fx["pos"].reshape(-1, 3).min(0), fx["pos"].reshape(-1, 3).max(0)
    assert np.all(fx["box"] >= lo - 1.0 - 1e-6) and np.all(fx["box"] <= hi + 1.0 + 1e-6)
    # dt = 0.02: a frame's displacement is its velocity times the step to within the SPH solver's own corrections
    d = np.linalg.norm(fx["pos"][1:] - fx["pos"][:-1], axis=-1).max()
    assert 0.03 < d < 0.12


@pytest.mark.parametrize("t", canyon.FRAMES)
def test_oracle_step_lands_nearer_the_next_frame(oracle, fx, weights, t):
    from oracle.model_ref import ModelRef
    ref = ModelRef(configs.LIQUID3D, weights)
    pos, vel = ref.step(canyon.inputs(fx, t))
    ratio, cos = canyon.score(fx, t, pos)
    assert ratio <= canyon.RATIO_BAR and cos >= canyon.COSINE_BAR, (t, ratio, cos)


def test_oracle_rollout_stays_near_the_frames(oracle, fx, weights):
    """Four free-running steps from frame 8 against frame 12 (tools/canyon.rollout_score): 0.30 measured, bar 0.4."""
    from oracle.model_ref import ModelRef
    ref = ModelRef(configs.LIQUID3D, weights)
    ratio = canyon.rollout_score(fx, lambda p, v: ref.step([p, v, None, None, fx["box"], fx["box_normals"]]))
    assert ratio <= canyon.ROLLOUT_BAR, ratio
    # ... and the bar has teeth where one step has none: out_scale x 0.5 (one step: 0.68 against 0.67) reads 0.47
    half = ModelRef(dict(configs.LIQUID3D, out_scale=[0.00390625] * 3), weights)
    assert canyon.rollout_score(fx, lambda p, v: half.step([p, v, None, None, fx["box"], fx["box_normals"]])) > canyon.ROLLOUT_BAR


def test_oracle_wrong_readings_are_told_apart(oracle, fx, weights):
    """The bars have teeth: three of the readings profiles/r06_reading_sweep.md separates, on one frame."""
    from oracle.model_ref import ModelRef
    from tools import reading_sweep as rs
    t = 10
    for ctx in (lambda: rs.filters_through(lambda f: f[::-1, ::-1, ::-1]),
                lambda: rs.filters_through(lambda f: f.transpose(2, 1, 0, 3, 4)),
                lambda: rs.conv_kwargs(align_corners=False)):
        with ctx():
            pos, _ = ModelRef(configs.LIQUID3D, weights).step(canyon.inputs(fx, t))
        ratio, cos = canyon.score(fx, t, pos)
        assert ratio > 1.0 and cos < 0.5, (ratio, cos)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["distance", "open3d"])
@pytest.mark.parametrize("t", canyon.FRAMES)
def test_hip_step_lands_nearer_the_next_frame(fx, weights, t, mode, monkeypatch):
    """The PRODUCT path (Simulator.step through libdmcf_hip.so) against the data -- no oracle in this test."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but torch.cuda.is_available() is False")
    from dmcf_amd import models
    from dmcf_amd.pipelines import Simulator
    from dmcf_amd.utils import tf_checkpoint as tc
    monkeypatch.setenv("DMCF_FRS_SET", mode)
    dev = torch.device("cuda:0")
    model = models.SymNet(**configs.LIQUID3D)
    tc.load_into_model(model, weights, device=dev)
    sim = Simulator(model, device="cuda")
    data = [None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in canyon.inputs(fx, t)]
    out = sim.step([data])[0]
    ratio, cos = canyon.score(fx, t, out[0].cpu().numpy())
    print(f"canyon frame {t} [{mode}]: ratio {ratio:.3f} cosine {cos:.3f}")
    assert ratio <= canyon.RATIO_BAR and cos >= canyon.COSINE_BAR, (t, ratio, cos)


@pytest.mark.gpu
def test_hip_rollout_stays_near_the_frames(fx, weights):
    """The PRODUCT path free-running for four steps from frame 8 against frame 12 -- no oracle in this test."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but torch.cuda.is_available() is False")
    from dmcf_amd import models
    from dmcf_amd.pipelines import Simulator
    from dmcf_amd.utils import tf_checkpoint as tc
    dev = torch.device("cuda:0")
    model = models.SymNet(**configs.LIQUID3D)
    tc.load_into_model(model, weights, device=dev)
    sim = Simulator(model, device="cuda")
    box, nrm = (torch.from_numpy(np.ascontiguousarray(fx[k])).to(dev) for k in ("box", "box_normals"))

    def step(p, v):
        out = sim.step([[torch.from_numpy(np.ascontiguousarray(p)).to(dev), torch.from_numpy(np.ascontiguousarray(v)).to(dev), None, None, box, nrm]])[0]
        return out[0].cpu().numpy(), out[1].cpu().numpy()
    ratio = canyon.rollout_score(fx, step)
    print(f"canyon 4-step rollout from frame 8: ratio {ratio:.3f}")
    assert ratio <= canyon.ROLLOUT_BAR, ratio
