"""Synthetic scenes for the parity tests and bench.py (SURVEY.md section 8d / BASELINE.md "Inputs").

The reference's datasets are not in the tree (download scripts, no network), so the benchmark
configurations use seeded synthetic stand-ins with the real network architectures.
"""
import numpy as np


def box_scene(side, h=0.05, jitter=0.1, vel_std=0.1, shell_layers=2, seed=0, dim=3, origin=(0.0, 0.0, 0.0), shell_refine=1):
    """Config 5 ("synthetic 3-D box"): a cube of side^3 fluid particles on a jittered lattice of spacing h
    (jitter U(-jitter*h, jitter*h), default_rng(seed)), velocities N(0, vel_std^2) (default_rng(seed+1)),
    enclosed by a closed shell of ``shell_layers`` lattice layers of boundary particles with inward normals.
    dim=2 gives the planar analogue (z = 0).  ``shell_refine`` = r: the shell's lattice has spacing h / r (the reference's own scenes
    sample their boundaries twice as densely as the fluid: canyon.msgpack.zst has 0.025 against 0.049).
    Returns dict(pos, vel, box, box_normals) float32."""
    rng = np.random.default_rng(seed)
    ax = (np.arange(side, dtype=np.float64) + 0.5) * h
    axes = [ax, ax, ax if dim == 3 else np.zeros(1)]
    pos = np.stack(np.meshgrid(*axes, indexing="ij"), -1).reshape(-1, 3)
    jit = rng.uniform(-jitter * h, jitter * h, size=pos.shape)
    if dim == 2:
        jit[:, 2] = 0
    pos = pos + jit
    vel = np.random.default_rng(seed + 1).normal(0.0, vel_std, size=pos.shape)
    if dim == 2:
        vel[:, 2] = 0
    L = shell_layers
    sr = side * int(shell_refine)
    bi = np.arange(-L, sr + L)
    gi = np.stack(np.meshgrid(*[bi if (dim == 3 or a < 2) else np.zeros(1, dtype=np.int64) for a in range(3)], indexing="ij"),
                  -1).reshape(-1, 3)
    nd = 3 if dim == 3 else 2
    keep = ((gi[:, :nd] < 0) | (gi[:, :nd] >= sr)).any(axis=1)
    gi = gi[keep]
    grid = (gi + 0.5) * (h / int(shell_refine))
    if dim == 2:
        grid[:, 2] = 0.0
    outside_lo = gi[:, :nd] < 0
    outside_hi = gi[:, :nd] >= sr
    box = grid
    normals = np.zeros_like(box)
    normals[:, :nd] = outside_lo.astype(np.float64) - outside_hi.astype(np.float64)
    normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    o = np.asarray(origin, dtype=np.float64)
    return dict(pos=(pos + o).astype(np.float32), vel=vel.astype(np.float32), box=(box + o).astype(np.float32),
                box_normals=normals.astype(np.float32))


def dam_break_scene(block=(50, 40, 50), tank=(100, 60, 50), h=0.05, jitter=0.1, shell_layers=2, seed=0):
    """BASELINE.json config 4 ("Liquid3d 3-D, ~100k particles"): a dam-break block of 50 x 40 x 50 = 100,000 fluid particles
    (lattice spacing h, jitter U(-jitter h, jitter h), default_rng(seed), at rest) standing in one corner of an OPEN tank of
    ``tank`` lattice cells -- floor and four walls of ``shell_layers`` boundary layers with inward normals, no lid; y is up
    (gravity (0, -9.81, 0) comes from the model).  Returns dict(pos, vel, box, box_normals) float32."""
    rng = np.random.default_rng(seed)
    axes = [(np.arange(n, dtype=np.float64) + 0.5) * h for n in block]
    pos = np.stack(np.meshgrid(*axes, indexing="ij"), -1).reshape(-1, 3)
    pos = pos + rng.uniform(-jitter * h, jitter * h, size=pos.shape)
    L = shell_layers
    g = [np.arange(-L, tank[0] + L), np.arange(-L, tank[1]), np.arange(-L, tank[2] + L)]  # no cells above the rim
    gi = np.stack(np.meshgrid(*g, indexing="ij"), -1).reshape(-1, 3)
    lo = gi < 0
    hi = gi >= np.asarray(tank)
    hi[:, 1] = False
    shell = (lo | hi).any(axis=1)
    box = (gi[shell] + 0.5) * h
    normals = lo[shell].astype(np.float64) - hi[shell].astype(np.float64)
    normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    return dict(pos=pos.astype(np.float32), vel=np.zeros_like(pos, dtype=np.float32), box=box.astype(np.float32),
                box_normals=normals.astype(np.float32))


def model_inputs(scene, device=None, grav=None):
    """[pos, vel, acc|None, feats|None, box, box_normals] as the Simulator feeds the model
    (pipelines/simulator.py:83-90).  numpy arrays, or torch tensors on ``device``."""
    acc = None
    if grav is not None:
        acc = np.broadcast_to(np.asarray(grav, dtype=np.float32), scene["pos"].shape).copy()
    data = [scene["pos"], scene["vel"], acc, None, scene["box"], scene["box_normals"]]
    if device is None:
        return data
    import torch
    return [None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(device) for x in data]


def random_weights(model_cfg, seed=0, lo=-0.05, hi=0.05, fluid_channels=None):
    """Seeded stand-in weights U(lo, hi) in the reference's checkpoint key layout, for the architectures whose
    trained blobs are absent (.MISSING_LARGE_BLOBS: WaterRamps, WBC-SPH).  Shapes follow the constructors
    (models/pbf_model.py:134-152, models/hrnet.py:39-67, models/sym_net.py:42-53, models/cconv.py:36-49)."""
    rng = np.random.default_rng(seed)
    c = dict(model_cfg)
    ks = list(c.get("kernel_size", [4, 4, 4]))
    name = c.get("name", "SymNet")
    lc = c["layer_channels"]
    w = {}

    def conv(key, cin, cout, kernel=ks, bias=True):
        w[key + "/kernel"] = rng.uniform(lo, hi, size=(*kernel, cin, cout)).astype(np.float32)
        if bias:
            w[key + "/bias"] = rng.uniform(lo, hi, size=(cout,)).astype(np.float32)

    def dense(key, cin, cout):
        w[key + "/kernel"] = rng.uniform(-0.3, 0.3, size=(cin, cout)).astype(np.float32)
        w[key + "/bias"] = rng.uniform(lo, hi, size=(cout,)).astype(np.float32)

    extra = (1 if c.get("dens_feats") else 0) + (1 if c.get("pres_feats") else 0)  # pbf_model.py:356-365
    n_fluid = (fluid_channels or (1 + (3 if c.get("use_vel", True) else 0) + (3 if c.get("use_acc", True) else 0))) + extra
    n_box = 1 + (3 if c.get("use_box_feats", True) else 0) + extra
    # dens_norm doubles the input of every layer that reads a scale with a density (hrnet.py:87-89)
    n_dens = len(c.get("dens_radius") or c.get("particle_radii", [0.05])) if c.get("dens_norm") else 0
    if name == "CConv":
        ch0 = lc[0]
        conv("model/fluid_convs", n_fluid, ch0)
        dense("model/fluid_dense", n_fluid, ch0)
        conv("model/obs_convs", n_box, ch0)
        dense("model/obs_dense", n_box, ch0)
        cin = 3 * ch0
        for i in range(1, len(lc)):
            conv(f"model/_all_convs/{i + 1}/1", cin, lc[i])
            dense(f"model/denses/{i - 1}", cin, lc[i])
            cin = lc[i]
        return w
    trunk = lc[:-1] if name == "SymNet" else lc
    ch0 = trunk[0][0][0]
    conv("model/fluid_convs", n_fluid, ch0)
    dense("model/fluid_dense", n_fluid, ch0)
    conv("model/obs_convs", n_box, ch0)
    dense("model/obs_dense", n_box, ch0)
    idx = 2
    prev = [3 * ch0]
    for i in range(1, len(trunk)):
        cur = []
        for j in range(len(trunk[i])):
            ch = trunk[i][j][0]
            for l in range(len(prev)):
                cin_l = prev[l] * (2 if l < n_dens else 1)
                conv(f"model/_all_convs/{idx}/1", cin_l, ch)
                idx += 1
                if l == j or c.get("voxel_size") is None:  # cross-scale Dense layers only exist on the FPS path (hrnet.py:100-113)
                    dense(f"model/denses/{i - 1}/{j}/0/{l}", cin_l, ch)
            cur.append(ch)
        prev = cur
    if name == "SymNet":
        sk = list(c.get("sym_kernel_size", [6, 6, 6]))
        sk[c.get("sym_axis", 2)] //= 2
        cin = prev[0]
        for i, ch in enumerate(lc[-1][-1]):
            conv(f"model/sym_convs/{i}", cin, ch, kernel=sk, bias=False)
            cin = ch
    return w


def box_block_scene(side, grid, rank, h=0.05, jitter=0.1, vel_std=0.1, shell_layers=2, seed=0):
    """The part of ONE (grid[0]*side) x (grid[1]*side) x (grid[2]*side) box (closed 2-layer shell around the whole box)
    that lies in the block of ``rank`` = (ix * grid[1] + iy) * grid[2] + iz (the rank order of
    dmcf_amd.parallel.BlockDecomposition): side^3 fluid particles with lattice indices [side*i, side*(i+1)) per axis,
    plus the pieces of the shell outside the block's outer faces.  Used by ``bench.py --gpus N`` (weak scaling: side^3
    fluid particles per GPU); each rank generates only its own part."""
    L = shell_layers
    grid = [int(g) for g in grid]
    # ``side``: one number (cubes) or three (the block's lattice cells per axis: strong scaling splits ONE box unevenly per axis)
    side = np.broadcast_to(np.asarray(side, dtype=np.int64), (3,))
    c = (rank // (grid[1] * grid[2]), (rank // grid[2]) % grid[1], rank % grid[2])
    rng = np.random.default_rng(seed + 2 * rank)
    axes = [(np.arange(side[k] * c[k], side[k] * (c[k] + 1)) + 0.5) * h for k in range(3)]
    pos = np.stack(np.meshgrid(*axes, indexing="ij"), -1).reshape(-1, 3)
    pos = pos + rng.uniform(-jitter * h, jitter * h, size=pos.shape)
    vel = np.random.default_rng(seed + 2 * rank + 1).normal(0.0, vel_std, size=pos.shape)
    g = []
    for k in range(3):
        lo = side[k] * c[k] - (L if c[k] == 0 else 0)
        hi = side[k] * (c[k] + 1) + (L if c[k] == grid[k] - 1 else 0)
        g.append(np.arange(lo, hi))
    gi = np.stack(np.meshgrid(*g, indexing="ij"), -1).reshape(-1, 3)
    total = np.array([side[0] * grid[0], side[1] * grid[1], side[2] * grid[2]])
    outside_lo = gi < 0
    outside_hi = gi >= total
    is_shell = (outside_lo | outside_hi).any(axis=1)
    box = (gi[is_shell] + 0.5) * h
    normals = outside_lo[is_shell].astype(np.float64) - outside_hi[is_shell].astype(np.float64)
    normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    return dict(pos=pos.astype(np.float32), vel=vel.astype(np.float32), box=box.astype(np.float32),
                box_normals=normals.astype(np.float32))


def box_slab_scene(side, world, rank, **kw):
    """:func:`box_block_scene` for cubes stacked along x."""
    return box_block_scene(side, [world, 1, 1], rank, **kw)
