"""GPU test of the sharded path with the real HIP kernels: N virtual ranks (threads, LocalComm) on one
MI355X against the single-rank result.  Real multi-process RCCL runs are the driver's (bench.py --gpus N)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _run_rank(comm, decomp, scene, steps, dev):
    from dmcf_amd import models, parallel
    from dmcf_amd.utils import tf_checkpoint as tc
    from tools import configs
    cfg = configs.LIQUID3D
    model = getattr(models, cfg["name"])(**cfg)  # one model object per virtual rank (layers keep per-call state)
    tc.load_into_model(model, dict(np.load(os.path.join(GOLDEN, "liquid3d_weights.npz"))), device=dev)
    sim = parallel.ShardedSimulator(model, comm, decomp)
    state = parallel.shard_scene(scene, decomp, comm.rank, dev)
    outs = []
    for _ in range(steps):
        state = sim.step(state)
        outs.append(sim.net_output.double().sum(0))
    return dict(gid=state["gid"].cpu().numpy(), pos=state["pos"].cpu().numpy(), vel=state["vel"].cpu().numpy(),
                exchanged=sim.exchanged_rows, out_sum=torch.stack(outs).cpu().numpy(), host_syncs=sim.host_syncs_last_step,
                out_abs=float(sim.net_output.double().abs().sum()))


def _assemble(parts, n):
    pos = np.zeros((n, 3), np.float32)
    vel = np.zeros((n, 3), np.float32)
    seen = np.zeros(n, bool)
    for p in parts:
        assert not seen[p["gid"]].any()
        seen[p["gid"]] = True
        pos[p["gid"]], vel[p["gid"]] = p["pos"], p["vel"]
    assert seen.all()
    return pos, vel


def _check(res, ref, n):
    pos1, vel1 = _assemble(ref, n)
    assert all(p["exchanged"] > 0 for p in res)
    pos, vel = _assemble(res, n)
    assert np.abs(pos - pos1).max() <= 1e-5 * np.abs(pos1).max()
    assert np.abs(vel - vel1).max() <= 2e-4 * np.abs(vel1).max()
    # momentum: the ASCC output summed over ALL ranks vanishes (ghost copies are bit-identical)
    total = sum(p["out_sum"] for p in res)
    scale = sum(p["out_abs"] for p in res)
    assert np.all(np.abs(total) <= 2e-5 * scale)


@pytest.mark.parametrize("world", [2, 4])
def test_virtual_ranks_match_single_rank_on_gpu(world, monkeypatch):
    from dmcf_amd import parallel
    monkeypatch.setenv("DMCF_SHARD_CHECK", "1")  # every derived ghost plan is compared with a directly built one
    from tools import scenes
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    parts = [scenes.box_slab_scene(10, 4, r, seed=7) for r in range(4)]  # 40 x 10 x 10 box, 4000 fluid particles
    scene = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
    n = scene["pos"].shape[0]
    ref = parallel.run_local_ranks(1, lambda c: _run_rank(c, parallel.SlabDecomposition(0, []), scene, 3, dev))
    decomp = parallel.SlabDecomposition.uniform(0, 0.0, 40 * 0.05, world)
    res = parallel.run_local_ranks(world, lambda c: _run_rank(c, decomp, scene, 3, dev))
    _check(res, ref, n)


def test_virtual_ranks_take_the_scatter_form_and_match_the_gather_form(monkeypatch):
    """Splat S (dmcf_cconv_scatter_forward) inside a SHARDED step: a rank's 24 -> 4 layer onto its owned s2 lattice points walks the
    transposed list over (owned + ghost) particles.  Two virtual ranks with the form forced on for these small blocks against one
    rank with the form off (splat F): positions within 1e-5, the scatter kernel seen on every rank."""
    from dmcf_amd import ops, parallel
    from dmcf_amd.utils import convolutions
    from tools import scenes
    dev = torch.device("cuda:0")
    parts = [scenes.box_slab_scene(14, 2, r, seed=11) for r in range(2)]  # 28 x 14 x 14 box
    scene = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
    n = scene["pos"].shape[0]
    monkeypatch.setenv("DMCF_SCATTER_CONV", "0")
    ref = parallel.run_local_ranks(1, lambda c: _run_rank(c, parallel.SlabDecomposition(0, []), scene, 3, dev))
    monkeypatch.setenv("DMCF_SCATTER_CONV", "1")
    monkeypatch.setattr(convolutions, "SCATTER_MIN_INPUTS", 256)
    ops.timer = ops.LaunchTimer()
    try:
        decomp = parallel.SlabDecomposition.uniform(0, 0.0, 28 * 0.05, 2)
        res = parallel.run_local_ranks(2, lambda c: _run_rank(c, decomp, scene, 3, dev))
        torch.cuda.synchronize()
        kernels = [m.get("kernel", "") for k, m, _ in ops.timer.results() if k == "cconv"]
    finally:
        ops.timer = None
    assert sum(k.startswith("cconv_sct_kernel") for k in kernels) >= 2 * 3, kernels[:40]
    _check(res, ref, n)


@pytest.mark.parametrize("grid", [[2, 2, 1], [2, 2, 2]])
def test_virtual_block_ranks_match_single_rank_on_gpu(grid, monkeypatch):
    """SURVEY.md section 8e's partitioning: 2x2x1 and 2x2x2 blocks of one 24^3 box (the pieces bench.py --gpus 4 / 8 gives
    its ranks, at side 12), real kernels, against one rank; face, edge and corner ghost peers; ASCC sum over ranks = 0."""
    from dmcf_amd import parallel
    monkeypatch.setenv("DMCF_SHARD_CHECK", "1")
    from tools import scenes
    dev = torch.device("cuda:0")
    world = grid[0] * grid[1] * grid[2]
    side = 12
    parts = [scenes.box_block_scene(side, grid, r, seed=3) for r in range(world)]
    scene = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
    n = scene["pos"].shape[0]
    ref = parallel.run_local_ranks(1, lambda c: _run_rank(c, parallel.SlabDecomposition(0, []), scene, 3, dev))
    decomp = parallel.BlockDecomposition.uniform([0.0, 0.0, 0.0], [g * side * 0.05 for g in grid], grid)
    # every rank's generated piece lies inside its block (what bench.py relies on with presharded=True)
    for r in range(world):
        parallel.shard_scene(parts[r], decomp, r, dev, presharded=True)
    res = parallel.run_local_ranks(world, lambda c: _run_rank(c, decomp, scene, 3, dev))
    _check(res, ref, n)
    ghosts = sum(p["exchanged"] for p in res)
    print(f"grid {grid}: {ghosts} ghost feature rows exchanged in 3 steps for {n} particles")


def test_fused_ghost_selection_gives_the_plans_of_the_host_form(monkeypatch):
    """The ghost plans of a step from csrc/ghost.hip (one count + one write per side and point set, all widths at once, the
    counts in one collective) against the torch form (wide plan + derived plans): the same particles to the bit, and fewer
    device -> host reads per step."""
    from dmcf_amd import parallel
    from tools import scenes
    dev = torch.device("cuda:0")
    grid = [2, 2, 1]
    parts = [scenes.box_block_scene(12, grid, r, seed=5) for r in range(4)]
    scene = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
    decomp = parallel.BlockDecomposition.uniform([0.0, 0.0, 0.0], [g * 12 * 0.05 for g in grid], grid)
    res = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("DMCF_SHARD_FUSED", fused)
        res[fused] = parallel.run_local_ranks(4, lambda c: _run_rank(c, decomp, scene, 3, dev))
    for a, b in zip(res["1"], res["0"]):
        assert np.array_equal(a["gid"], b["gid"]) and np.array_equal(a["pos"], b["pos"]) and np.array_equal(a["vel"], b["vel"])
        assert a["exchanged"] == b["exchanged"]
        assert a["host_syncs"] < b["host_syncs"], (a["host_syncs"], b["host_syncs"])
    print("host reads per step, fused / host form:", res["1"][0]["host_syncs"], res["0"][0]["host_syncs"])


def _proc_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from dmcf_amd import parallel
    from tools import scenes
    dev = torch.device("cuda:0")  # both processes share the one GPU of the box; the collectives run over gloo
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        grid = [2, 1, 1]
        scene = scenes.box_block_scene(12, grid, rank, seed=3)
        decomp = parallel.BlockDecomposition.uniform([0.0, 0.0, 0.0], [g * 12 * 0.05 for g in grid], grid)
        comm = parallel.TorchDistComm()
        from dmcf_amd import models
        from dmcf_amd.utils import tf_checkpoint as tc
        from tools import configs
        cfg = configs.LIQUID3D
        model = getattr(models, cfg["name"])(**cfg)
        tc.load_into_model(model, dict(np.load(os.path.join(GOLDEN, "liquid3d_weights.npz"))), device=dev)
        sim = parallel.ShardedSimulator(model, comm, decomp)
        state = parallel.shard_scene(scene, decomp, rank, dev, presharded=True)
        state["gid"] = state["gid"] + rank * scene["pos"].shape[0]
        for _ in range(3):
            state = sim.step(state)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), gid=state["gid"].cpu().numpy(), pos=state["pos"].cpu().numpy(),
                 vel=state["vel"].cpu().numpy(), exchanged=sim.exchanged_rows)
    finally:
        dist.destroy_process_group()


def test_two_processes_one_gpu_match_single_rank(tmp_path):
    """Two REAL processes (torch.distributed, TorchDistComm, the bench's pre-sharded scene pieces) with the real kernels:
    they share the box's single GPU and talk over gloo -- everything of the N > 1 bench path except RCCL itself."""
    import torch.multiprocessing as mp
    from dmcf_amd import parallel
    from tools import scenes
    dev = torch.device("cuda:0")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_proc_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = [dict(np.load(os.path.join(tmp_path, f"rank{r}.npz"))) for r in range(2)]
    parts = [scenes.box_block_scene(12, [2, 1, 1], r, seed=3) for r in range(2)]
    scene = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
    n = scene["pos"].shape[0]
    ref = parallel.run_local_ranks(1, lambda c: _run_rank(c, parallel.SlabDecomposition(0, []), scene, 3, dev))
    pos1, vel1 = _assemble(ref, n)
    assert all(int(p["exchanged"]) > 0 for p in res)
    pos, vel = _assemble(res, n)
    assert np.abs(pos - pos1).max() <= 1e-5 * np.abs(pos1).max()
    assert np.abs(vel - vel1).max() <= 2e-4 * np.abs(vel1).max()


def test_sharded_bench_over_rccl_world1(tmp_path):
    """The N > 1 code path of bench.py -- ShardedSimulator, BlockDecomposition, the pre-sharded scene pieces, TorchDistComm over
    the "nccl" (= RCCL) backend -- at world size 1, the most a 1-GPU box can run of it: started the way the driver starts a
    rank (torch.distributed.run), one JSON line, the same particle count in and out."""
    import json
    import subprocess
    import sys
    # (DMCF_SHARD_FORCE_COMM: the feature exchanges -- asynchronous all-to-all-v over RCCL, started when a layer begins -- run
    # although a single rank has no ghosts)
    env = dict(os.environ, DMCF_BENCH_SHARDED="1", DMCF_SHARD_FORCE_COMM="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + os.getpid() % 2000), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--side", "24",
           "--steps", "2", "--warmup", "2", "--cpu-side", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and "1x1x1 blocks" in d["config"]["parallelism"] and d["value"] > 0
    assert d["roofline_groups"]["neighbour_list"]["launches"] > 0
    # what the first real multi-GPU run will be read by: per-rank ghost / migration rows, device -> host reads, exposed waits
    pr = d["per_rank"]
    assert len(pr) == 1 and pr[0]["rank"] == 0 and pr[0]["fluid_particles"] == 24 ** 3
    assert {"ghost_rows_per_step", "migrated_rows_per_step", "host_syncs_per_step", "exposed_exchange_wait_ms_per_step",
            "rank_seconds"} <= set(pr[0])
    assert pr[0]["exposed_exchange_wait_ms_per_step"] >= 0 and pr[0]["host_syncs_per_step"] > 0
    assert d["scene_state"]["fluid_outside_shell_at_end"] == 0 and "weak scaling" in d["config"]["parallelism"]
