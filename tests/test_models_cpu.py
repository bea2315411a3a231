"""CPU tests of the host logic: configs, checkpoint reader, model construction (no GPU compute)."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def test_config_dicts_match_reference_yaml():
    if not os.path.isdir(REF):
        pytest.skip("reference tree not present (GPU box)")
    from dmcf_amd.utils.config import Config
    from tools import configs
    for name, d in configs.BY_NAME.items():
        cfg = Config.load_from_file(os.path.join(REF, "configs", name + ".yml"))
        for k, v in d.items():
            assert cfg.model[k] == v, (name, k)


def test_config_cli_overrides():
    from dmcf_amd.utils.config import Config
    cfg = Config(dict(dataset=dict(dataset_path=None), model=dict(timestep=0.02, use_acc=False, strides=[1, 2]),
                      pipeline=dict(version="3d")))
    d, p, m = Config.merge_cfg_file(cfg, dict(dataset_path="/data", ckpt_path="ck"),
                                    {"model.timestep": "0.01", "model.use_acc": "true", "model.strides": "[1, 2, 4]"})
    assert d["dataset_path"] == "/data" and m["ckpt_path"] == "ck"
    assert m["timestep"] == 0.01 and m["use_acc"] is True and m["strides"] == [1, 2, 4]


@pytest.mark.parametrize("name,n_convs", [("Liquid3d", 18), ("WaterRamps", 27), ("WBC-SPH", 43)])
def test_architecture_matches_checkpoint_index(name, n_convs):
    """Golden check against the reference's own checkpoint indices (tests/golden/ckpt_shapes.json): the layer
    list built from the config has exactly the variables / shapes the shipped checkpoints hold."""
    from dmcf_amd import models
    from dmcf_amd.utils import tf_checkpoint as tc
    from tools import configs, scenes
    shapes = json.load(open(os.path.join(GOLDEN, "ckpt_shapes.json")))[name]
    cfg = configs.BY_NAME[name]
    model = getattr(models, cfg["name"])(**cfg)
    assert len(model._all_convs) == n_convs
    w = scenes.random_weights(cfg)
    assert {k: list(v.shape) for k, v in w.items()} == shapes  # same keys, same shapes
    assert tc.load_into_model(model, w, device="cpu") == len([k for k in shapes if k.endswith("/kernel")])


def test_tf_checkpoint_reader_on_reference_blob():
    if not os.path.isdir(REF):
        pytest.skip("reference tree not present (GPU box)")
    from dmcf_amd.utils import tf_checkpoint as tc
    w = tc.load_checkpoint(os.path.join(REF, "checkpoints/Liquid3d/ckpt"))
    model_vars = {k: v for k, v in w.items() if k.startswith("model/")}
    assert sum(v.size for v in model_vars.values()) == 275084
    assert all(np.isfinite(v).all() for v in model_vars.values())
    fix = np.load(os.path.join(GOLDEN, "liquid3d_weights.npz"))
    assert sorted(fix.files) == sorted(model_vars)
    for k in fix.files:
        np.testing.assert_array_equal(fix[k], model_vars[k])
    assert int(w["optimizer/iter"]) == 51000


def test_checkpoint_epoch_rule():
    """base_pipeline.py:171-185: the newest checkpoint of a manager directory, 'ckpt-<n>', continues at epoch
    (n - 1) * save_ckpt_freq + 1 (an explicit ckpt_path is epoch 0: Simulator.load_ckpt)."""
    from dmcf_amd.utils import tf_checkpoint as tc
    assert tc.checkpoint_epoch("/logs/checkpoint/ckpt-12") == 12
    assert tc.checkpoint_epoch("/logs/checkpoint/ckpt-12", save_ckpt_freq=5) == 56
    assert tc.checkpoint_epoch("/logs/checkpoint/ckpt-1", save_ckpt_freq=5) == 1
    assert tc.checkpoint_epoch("/checkpoints/Liquid3d/ckpt") == 0


def test_box_scene_generator():
    from tools import scenes
    s = scenes.box_scene(10)
    assert s["pos"].shape == (1000, 3) and s["box"].shape == (14 ** 3 - 10 ** 3, 3)
    np.testing.assert_allclose(np.linalg.norm(s["box_normals"], axis=1), 1, atol=1e-6)
    s2 = scenes.box_scene(10, dim=2)
    assert s2["pos"].shape == (100, 3) and np.all(s2["pos"][:, 2] == 0) and s2["box"].shape == (14 ** 2 - 100, 3)


def test_window_functions_match_oracle(oracle):
    import torch
    from dmcf_amd.utils.tools.losses import get_window_func
    q = np.linspace(0, 1.2, 50).astype(np.float32)
    for typ in ("poly6", "cubic", "linear", "peak", "cubic_grad"):
        w = get_window_func(typ)(torch.from_numpy(q)).numpy()
        np.testing.assert_allclose(w, oracle.window(typ, q), atol=1e-6)
    assert get_window_func(None) is None
    with pytest.raises(NotImplementedError):
        get_window_func("gauss")


def test_grid_pos_matches_oracle(oracle):
    import torch
    from dmcf_amd.utils.tools.losses import grid_pos, get_dilated_pos
    rng = np.random.default_rng(0)
    pos = rng.uniform(-1, 1, size=(3000, 3)).astype(np.float32)
    for vs, cen in (([0.1, 0.1, 0.1], True), ([0.1, 0.1, 0.1], False), ([0.05, 0.05, 0.0], True), ([0.0, 0.02, 0.0], True)):
        p = pos.copy()
        for a in range(3):
            if vs[a] == 0:
                p[:, a] = 0
        g = grid_pos(torch.from_numpy(p), vs, centralize=cen).numpy()
        ref = oracle.grid_pos(p, vs, centralize=cen)
        assert g.shape == ref.shape
        # same lattice points in the same (first occurrence) order; centre may differ by float rounding of the mean
        np.testing.assert_allclose(g, ref, atol=2e-6)
    d, cnt, idx = get_dilated_pos(torch.from_numpy(pos), [1, 2, 4], voxel_size=[0.05, 0.05, 0.05], centralize=True)
    assert cnt[0] == 3000 and cnt[1] > cnt[2] and idx == [None]


def test_density_feature_flags_host_logic(oracle, monkeypatch):
    """dens_feats / pres_feats / dens_norm (models/pbf_model.py:351-365,421-431; models/hrnet.py:87-89) wired the way
    the reference wires them: the model on CPU tensors with the operators routed through the oracle (tests/shims.py)
    equals the numpy restatement of the model."""
    import torch
    import shims
    from oracle.model_ref import ModelRef
    from dmcf_amd import models
    from dmcf_amd.utils import tf_checkpoint as tc
    from tools import configs, scenes
    shims.install(monkeypatch)
    cfg = dict(configs.WATERRAMPS, dens_feats=True, pres_feats=True, dens_norm=True, window_dens="poly6", rest_dens=12.0)
    w = scenes.random_weights(cfg, seed=4)
    scene = scenes.box_scene(14, h=0.005, dim=2, origin=(-0.03, -0.03, 0.0))
    model = getattr(models, cfg["name"])(**cfg)
    tc.load_into_model(model, w, device="cpu")
    ref = ModelRef(cfg, w)
    data_np = scenes.model_inputs(scene)
    data_t = scenes.model_inputs(scene, device="cpu")
    pos_ref, vel_ref = ref.step(data_np)
    pos, vel = model(data_t, training=False)[:2]
    assert np.abs(pos.numpy() - pos_ref).max() <= 1e-6 * np.abs(pos_ref).max()
    assert np.abs(model.pos_correction.numpy() - ref.pos_correction).max() <= 1e-4 * np.abs(ref.pos_correction).max()
    assert ref.dens is not None and len(ref.dens) == len(cfg["particle_radii"])


def test_scene_file_io_roundtrip_and_reference_fixture(tmp_path):
    """datasets: the reference's scene format (zstd frame > msgpack list of frame dicts > msgpack-numpy arrays).
    tests/golden/canyon_crop.msgpack.zst holds frames of the reference's own canyon scene (make_fixtures.py)."""
    from dmcf_amd.datasets import Dataset, get_rollout, read_scene, write_results, write_results_npz, write_scene
    frames = read_scene(os.path.join(ROOT, "tests", "golden", "canyon_crop.msgpack.zst"))
    assert len(frames) == 3 and frames[0]["pos"].shape == (1280, 3) and frames[0]["pos"].dtype == np.float32
    assert frames[0]["box"].shape == (10006, 3) and frames[0]["box_normals"].shape == (10006, 3)
    assert "box" not in frames[1] and int(frames[2]["frame_id"]) == 2 and frames[0]["scene_id"] == "sim_canyon"
    np.testing.assert_allclose(np.linalg.norm(frames[0]["box_normals"], axis=1), 1.0, atol=1e-4)
    # write -> read round trip, bit exact, scalars and strings included
    p = str(tmp_path / "a.msgpack.zst")
    write_scene(p, frames, level=3)
    again = read_scene(p)
    for a, b in zip(frames, again):
        assert a.keys() == b.keys()
        for k in a:
            assert np.array_equal(a[k], b[k]) and type(a[k]) is type(b[k])
    # directory data set + get_rollout (dataset_reader_physics.py:410-456) with the input transform
    ds = Dataset(dataset_path=str(tmp_path))
    assert len(ds) == 1
    r = get_rollout(ds, translate=[0.5, 0.0, 0.0], scale=[1.0, 1.0, 1.0])[0]
    assert r["pos"].shape == (3, 1280, 3) and r["box"].shape == (3, 10006, 3) and list(r["frame_id"]) == [0, 1, 2]
    np.testing.assert_array_equal(r["pos"][1], frames[1]["pos"] + np.float32([0.5, 0, 0]))
    assert get_rollout(ds, time_start=1, time_end=2)[0]["pos"].shape[0] == 1
    # result writers: the reference's HDF5 file (h5py or, without it, the built-in writer: tests/test_hdf5_writer.py reads it
    # back with the HDF5 C library), and an .npz with the same content
    out = [(r["pos"], {"name": "pred", "type": "PARTICLE"}), (frames[0]["box"], {"name": "bnd", "type": "PARTICLE"})]
    write_results_npz(str(tmp_path / "r.npz"), "SymNet", out)
    z = np.load(str(tmp_path / "r.npz"))
    assert z["SymNet/pred"].shape == (3, 1280, 3) and str(z["SymNet/bnd.type"]) == "PARTICLE"
    write_results(str(tmp_path / "r.hdf5"), "SymNet", out)
    import test_hdf5_writer as h5t
    got = h5t._walk(str(tmp_path / "r.hdf5"))["SymNet"]
    np.testing.assert_array_equal(got["pred"], r["pos"])
    np.testing.assert_array_equal(got["bnd"], frames[0]["box"])


def test_fps_multiscale_host_logic(oracle, monkeypatch):
    """voxel_size: None multi-scale (losses.py:274-282) + the cross-scale Dense branch of HRNet (hrnet.py:100-113):
    the model on CPU tensors with oracle-backed operators equals the numpy restatement."""
    import shims
    from oracle.model_ref import ModelRef
    from dmcf_amd import models
    from dmcf_amd.utils import tf_checkpoint as tc
    from tools import configs, scenes
    shims.install(monkeypatch)
    cfg = dict(configs.WATERRAMPS, voxel_size=None, centralize=False)
    w = scenes.random_weights(cfg, seed=6)
    scene = scenes.box_scene(12, h=0.005, dim=2, origin=(-0.03, -0.03, 0.0))
    model = getattr(models, cfg["name"])(**cfg)
    tc.load_into_model(model, w, device="cpu")
    ref = ModelRef(cfg, w)
    pos_ref, vel_ref = ref.step(scenes.model_inputs(scene))
    pos, vel = model(scenes.model_inputs(scene, device="cpu"), training=False)[:2]
    assert np.abs(pos.numpy() - pos_ref).max() <= 1e-6 * np.abs(pos_ref).max()
    assert np.abs(model.pos_correction.numpy() - ref.pos_correction).max() <= 1e-4 * np.abs(ref.pos_correction).max()
    assert [len(i) for i in ref.fps_idx[1:]] == [len(model.all_pos) // 2, len(model.all_pos) // 4]


def test_run_pipeline_arguments_and_dataset_group(tmp_path):
    """run_pipeline.py:12-52 (flags + --section.key overrides) and DatasetGroup (dataset_reader_physics.py:85-142) without
    a GPU: the test split resolves <path>/test or <path>; generators and the train split raise."""
    import yaml
    from dmcf_amd import run_pipeline
    from dmcf_amd.datasets import DatasetGroup, read_scene
    from dmcf_amd.utils.config import Config
    from tools import configs
    cfg = dict(dataset=dict(name="D"), model=dict(configs.LIQUID3D), pipeline=dict(name="Simulator", version="v0"))
    yml = tmp_path / "c.yml"
    yml.write_text(yaml.safe_dump(cfg))
    args, extra = run_pipeline.parse_args(["-c", str(yml), "--split", "test", "--dataset_path", GOLDEN, "--model.timestep", "0.01",
                                           "--ckpt_path", "/x/ckpt"])
    assert extra == {"model.timestep": "0.01"} and args.split == "test"
    d, p, m = Config.merge_cfg_file(Config.load_from_file(str(yml)), args, extra)
    assert m["timestep"] == 0.01 and m["ckpt_path"] == "/x/ckpt" and d["dataset_path"] == GOLDEN and p["split"] == "test"
    g = DatasetGroup(**d, split="test")
    assert len(g.test) == 1 and g.valid is g.test and len(g.test[0]) == 3  # the canyon crop: one scene file, three frames
    g2 = DatasetGroup(name="x", data=[read_scene(os.path.join(GOLDEN, "canyon_crop.msgpack.zst"))])
    assert len(g2.test) == 1
    with pytest.raises(NotImplementedError):
        DatasetGroup(name="x", type="column", split="test")
    with pytest.raises(NotImplementedError):
        DatasetGroup(name="x", dataset_path=GOLDEN, split="train")
    with pytest.raises(NotImplementedError):
        run_pipeline.main(["-c", str(yml), "--split", "train"])


def test_neighbour_list_estimates_are_keyed_by_what_a_search_is():
    """utils/convolutions.py: the longest-row estimates a step hands to the next belong to a search's CLASS (radius, size class of
    both point sets, flags), not to its position in the step -- a layer that falls back from the lattice form adds a search
    in the middle of the sequence and must not shift every later estimate."""
    import torch
    from dmcf_amd.utils import convolutions as cv

    class Frs:
        def __init__(self, ignore=False, dist=True):
            self.ignore_query_point, self.return_distances = ignore, dist

    def key(n_points, n_queries, radius, **kw):
        return cv._hint_key(Frs(**kw), torch.empty(n_points, 3), torch.empty(n_queries, 3), radius)

    assert key(1_124_864, 1_157_625, 0.2) == key(1_130_000, 1_150_000, 0.2)  # particle counts drift by a few per cent per step
    assert key(1_124_864, 1_157_625, 0.2) != key(1_124_864, 1_157_625, 0.4)
    assert key(1_124_864, 151_686, 0.4) != key(151_686, 1_124_864, 0.4)       # s0 -> s2 is not s2 -> s0
    assert key(1000, 1000, 0.1, ignore=True) != key(1000, 1000, 0.1) != key(1000, 1000, 0.1, dist=False)
    assert key(0, 0, 0.1) == key(1, 1, 0.1)                                   # (empty sets have a class too)
    view = cv._HintView({key(1000, 1000, 0.1): 35, key(1000, 2000, 0.2): 286})
    assert sorted(view) == [35, 286] and len(view) == 2
    view.clear()
    assert list(view) == [] and len(view) == 0
    assert cv.row_stride(35) >= 35 + 8 and cv.row_stride(35) % 8 == 0 and cv.row_stride(2964) >= 2964 * 5 // 4

