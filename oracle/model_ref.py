"""numpy restatement of one DMCF model step (test infrastructure only; parity unpinned).

Follows the reference's model code line by line on numpy float32, calling the C oracle for the Open3D
operators.  It exists (a) as the checker for the whole-step parity tests and (b) as the timed CPU
baseline of bench.py (``cpu_baseline.kind = "port"``).  Paths relative to /root/reference.

  BaseModel.call            models/base_model.py:23-29
  PBFNet.transform/...      models/pbf_model.py:234-301 (integrate, transforms), :303-438, :440-489
  HRNet.forward             models/hrnet.py:69-133
  SymNet.forward            models/sym_net.py:55-69
  CConv.forward             models/cconv.py:51-69
"""
import numpy as np

from . import oracle as O

f32 = np.float32


def _relu(x):
    return np.maximum(x, f32(0))


def align_vector(v0, v1):
    """models/pbf_model.py:12-28."""
    v0n = v0 / (np.linalg.norm(v0) + f32(1e-9))
    v1n = v1 / (np.linalg.norm(v1) + f32(1e-9))
    v = np.cross(v0n, v1n)
    c = np.dot(v0n, v1n)
    s = np.linalg.norm(v)
    if s < 1e-6:
        return (np.eye(3) * (-1.0 if c < 0 else 1.0)).astype(f32)
    vx = np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]], dtype=f32)
    return (np.eye(3, dtype=f32) + vx + (vx @ vx) / (1 + c)).astype(f32)


class ModelRef:
    """One object per (config, weights).  ``weights``: {reference checkpoint key: array}; conv i of the
    creation order is looked up as model/_all_convs/<i>/1 or under its attribute alias."""

    def __init__(self, model_cfg, weights, f64=False):
        c = dict(model_cfg)
        self.kind = c.get("name", "SymNet")
        self.cfg = c
        self.w = weights
        self.f64 = f64
        self.kernel_size = c.get("kernel_size", [4, 4, 4])
        self.strides = c.get("strides", [1])
        self.particle_radii = c.get("particle_radii", [0.05])
        self.coordinate_mapping = c.get("coordinate_mapping", "ball_to_cube_volume_preserving")
        self.interpolation = c.get("interpolation", "linear")
        self.window = c.get("window")
        self.window_sym = c.get("window_sym")
        self.ignore_query_points = c.get("ignore_query_points", False)
        self.grav = c.get("grav", -9.81)
        self.transformation = c.get("transformation") or {}
        self.timestep = c.get("timestep", 0.01)
        self.use_vel = c.get("use_vel", True)
        self.use_acc = c.get("use_acc", True)
        self.use_box_feats = c.get("use_box_feats", True)
        self.use_bnds = c.get("use_bnds", True)
        self.dens_feats = c.get("dens_feats", False)
        self.pres_feats = c.get("pres_feats", False)
        self.dens_norm = c.get("dens_norm", False)
        self.window_dens = c.get("window_dens")
        self.dens_radius = c.get("dens_radius") or c.get("particle_radii", [0.05])
        self.rest_dens = c.get("rest_dens", 3.5)
        self.stiffness = c.get("stiffness", 20.0)
        self.voxel_size = c.get("voxel_size")
        self.centralize = c.get("centralize", False)
        self.out_scale = np.asarray(c.get("out_scale", [0.01, 0.01, 0.01]), dtype=f32)
        self.sample_pad = c.get("sample_pad", 0)
        self.sample_hyst = c.get("sample_hyst", 0.1)
        self.part_scale = f32(c.get("part_scale", 1.0))
        self.add_merge = c.get("add_merge", False)
        self.sym_kernel_size = c.get("sym_kernel_size", [6, 6, 6])
        self.sym_axis = c.get("sym_axis", 2)
        lc = c.get("layer_channels")
        self.layer_channels = lc
        self._conv_count = 0
        self._alias = {}
        self.nns_cache = {}
        self.pairs = 0
        self.record = None  # set to [] (or a callable) to record every CConv call (inputs + output) of the next step

    # ---- weights --------------------------------------------------------------------------------
    def _conv_weights(self, index, alias=None):
        for prefix in ([alias] if alias else []) + [f"model/_all_convs/{index}/1"]:
            if prefix + "/kernel" in self.w:
                return self.w[prefix + "/kernel"], self.w.get(prefix + "/bias")
        raise KeyError(f"conv {index} ({alias}) not in weights")

    def _dense(self, key, x):
        return (x @ self.w[key + "/kernel"] + self.w[key + "/bias"]).astype(f32)

    def _cconv(self, index, alias, feats, inp_pos, out_pos, extent, window, ignore=False, symmetric=False,
               kernel_size=None):
        kernel, bias = self._conv_weights(index, alias)
        key = (id(inp_pos), id(out_pos), float(extent), bool(ignore))
        radius = f32(0.5) * f32(extent)
        if key not in self.nns_cache:  # identical result to searching again; saves oracle time
            self.nns_cache[key] = O.fixed_radius_search(inp_pos, out_pos, radius, ignore)
        nns = self.nns_cache[key]
        self.pairs += int(nns[0].shape[0])
        conv = O.ContinuousConvRef(kernel, bias=None if symmetric else bias, window_function=window,
                                   ignore_query_points=ignore, symmetric=symmetric, sym_axis=self.sym_axis,
                                   normalize=False, align_corners=True, coordinate_mapping=self.coordinate_mapping,
                                   interpolation=self.interpolation, f64=self.f64)
        out = conv(feats, inp_pos, out_pos, f32(extent), nns=nns)
        self.last_nns = nns
        if self.record is not None:
            rec = dict(index=index, kernel=kernel, bias=None if symmetric else bias, feats=feats,
                       inp_pos=inp_pos, out_pos=out_pos, extent=float(extent), window=window,
                       ignore=ignore, symmetric=symmetric, nns=nns, out=out)
            if callable(self.record):  # (a diagnostic that replays the call right away instead of keeping 1M-particle lists)
                self.record(rec)
            else:
                self.record.append(rec)
        return out

    # ---- stages ---------------------------------------------------------------------------------
    def integrate_pos_vel(self, pos1, vel1, acc1=None):  # pbf_model.py:234-240
        dt = f32(self.timestep)
        a = acc1 if acc1 is not None else np.array([0, self.grav, 0], dtype=f32)
        vel2 = (vel1 + dt * a).astype(f32)
        pos2 = (pos1 + dt * vel2).astype(f32)
        return pos2, vel2

    def transform(self, data):  # pbf_model.py:252-280
        pos, vel, acc, feats, box, bfeats = data
        if "translate" in self.transformation:
            t = np.asarray(self.transformation["translate"], dtype=f32)
            pos, box = pos + t, box + t
        if "scale" in self.transformation:
            s = np.asarray(self.transformation["scale"], dtype=f32)
            pos, box, vel = pos * s, box * s, vel * s
            if acc is not None:
                acc = acc * s
        if "grav_eqvar" in self.transformation:
            g = np.asarray(self.transformation["grav_eqvar"], dtype=f32)
            self.R = align_vector(g, acc[0])
            pos, vel, acc, box, bfeats = (x @ self.R for x in (pos, vel, acc, box, bfeats))
        return [pos, vel, acc, feats, box, bfeats]

    def inv_transform(self, prev):  # pbf_model.py:282-301
        pos, vel = prev
        if "grav_eqvar" in self.transformation:
            R = self.R.T
            pos, vel = pos @ R, vel @ R
        if "scale" in self.transformation:
            s = np.maximum(np.asarray(self.transformation["scale"], dtype=f32), f32(1e-5))
            pos, vel = pos / s, vel / s
        if "translate" in self.transformation:
            pos = pos - np.asarray(self.transformation["translate"], dtype=f32)
        return pos.astype(f32), vel.astype(f32)

    def preprocess(self, data):  # pbf_model.py:303-438
        _pos, _vel, acc, feats, box, bfeats = data
        pos, vel = self.integrate_pos_vel(_pos, _vel, acc)
        filter_extent = np.asarray(self.particle_radii, dtype=f32) * f32(2)
        fltr = np.all([box >= pos.min(axis=0) - filter_extent[-1], box <= pos.max(axis=0) + filter_extent[-1]],
                      axis=(0, 2))
        box = np.ascontiguousarray(box[fltr])
        bfeats = np.ascontiguousarray(bfeats[fltr])
        fluid_feats = [np.ones_like(pos[:, :1])]
        if self.use_vel:
            fluid_feats.append(vel)
        if self.use_acc:
            fluid_feats.append(acc)
        box_feats = [np.ones_like(box[:, :1])]
        if self.use_box_feats:
            box_feats.append(bfeats)
        all_pos = np.ascontiguousarray(np.concatenate([pos, box], axis=0))
        self.all_pos = all_pos
        dens = None
        if self.dens_feats or self.dens_norm or self.pres_feats:  # pbf_model.py:351-365
            dens = O.compute_density(all_pos, all_pos, self.dens_radius[0], self.window_dens)
            n = pos.shape[0]
            if self.dens_feats:
                fluid_feats.append(dens[:n, None])
                box_feats.append(dens[n:, None])
            if self.pres_feats:
                pres = O.compute_pressure(dens, self.rest_dens, self.stiffness)
                fluid_feats.append(pres[:n, None])
                box_feats.append(pres[n:, None])
        fluid_feats = np.concatenate(fluid_feats, axis=-1).astype(f32)
        box_feats = np.concatenate(box_feats, axis=-1).astype(f32)
        # get_cconv (pbf_model.py:208-209): ignore_query_points=None -> the model-level flag, also for the input convs
        ans_conv = self._cconv(0, "model/fluid_convs", fluid_feats * self.part_scale, pos, all_pos, filter_extent[0],
                               self.window, ignore=self.ignore_query_points)
        self.fluid_nns = self.last_nns
        ans_dense = self._dense("model/fluid_dense", fluid_feats)
        ans_obs = self._cconv(1, "model/obs_convs", box_feats * self.part_scale, box, all_pos, filter_extent[0],
                              self.window, ignore=self.ignore_query_points)
        ans_dense_obs = self._dense("model/obs_dense", box_feats)
        ans_dense = np.concatenate([ans_dense, ans_dense_obs], axis=0)
        feats_out = np.concatenate([ans_conv, ans_obs, ans_dense], axis=-1).astype(f32)
        base = all_pos if self.use_bnds else pos
        if self.voxel_size is not None:
            dilated = O.get_dilated_pos(base, self.strides, self.voxel_size, self.centralize, self.sample_pad,
                                        self.sample_hyst)
        else:
            dilated, self.fps_idx = O.get_dilated_pos_fps(base, self.strides)  # losses.py:274-282
        dilated = [np.ascontiguousarray(d) for d in dilated]
        dilated[0] = base  # keep identity for the neighbour cache
        self.dilated_pos = dilated  # (diagnostics: tools/diag_degraded.py compares the lattices point by point)
        self.dens = None
        if self.dens_norm:  # pbf_model.py:421-431
            self.dens = [(dens if self.use_bnds else dens[:pos.shape[0]])[:, None].astype(f32)]
            for scale in range(1, len(self.dens_radius)):
                d = O.point_sampling(self.dens[-1], dilated[scale - 1], dilated[scale], self.dens_radius[scale],
                                     self.window_dens, normalize=True, f64=self.f64)
                self.dens.append(np.maximum(d, f32(1e-2)).astype(f32))
        self._conv_count = 2
        return dilated, feats_out

    def hrnet_forward(self, pos, feats, layer_channels):  # hrnet.py:39-67 (indices), :69-133
        filter_extent = np.asarray(self.particle_radii, dtype=f32) * f32(2)
        if not self.use_bnds:
            feats = feats[:pos[0].shape[0]]
        ans_convs = [[feats]]
        for i in range(1, len(layer_channels)):
            layer = i - 1
            ans = []
            for scale in range(len(layer_channels[i])):
                assert len(layer_channels[i][scale]) == 1, "k > 0 sub-layers are not used by shipped configs"
                importance = self.part_scale if scale == 0 else f32(1.0)
                inp = []
                for inp_scale in range(len(ans_convs[-1])):
                    f = _relu(ans_convs[-1][inp_scale])
                    if self.dens_norm and self.dens is not None and inp_scale < len(self.dens):  # hrnet.py:87-89
                        f = np.concatenate([f, f / self.dens[inp_scale] ** 2], axis=-1).astype(f32)
                    ext = filter_extent[max(inp_scale, scale)]
                    index = self._conv_count
                    self._conv_count += 1
                    ignore = self.ignore_query_points and (scale == inp_scale)
                    ans_conv = self._cconv(index, None, (f * importance).astype(f32), pos[inp_scale], pos[scale], ext,
                                           self.window, ignore=ignore)
                    if scale == inp_scale:
                        ans_conv = ans_conv + self._dense(f"model/denses/{layer}/{scale}/0/{inp_scale}", f)
                        if ans_conv.shape[-1] == ans_convs[-1][scale].shape[-1]:
                            ans_conv = ans_conv + ans_convs[-1][scale]
                    elif self.voxel_size is None:  # hrnet.py:100-113
                        idx = self.fps_idx
                        if scale > inp_scale:
                            g = f
                            for s_ in range(inp_scale, scale):
                                g = g[idx[s_ + 1]]
                            ans_conv = ans_conv + self._dense(f"model/denses/{layer}/{scale}/0/{inp_scale}", g)
                        else:
                            ind = idx[scale + 1]
                            for s_ in range(scale + 1, inp_scale):
                                ind = ind[idx[s_ + 1]]
                            ans_conv = ans_conv.copy()
                            np.add.at(ans_conv, ind, self._dense(f"model/denses/{layer}/{scale}/0/{inp_scale}", f))
                    inp.append(ans_conv.astype(f32))
                if self.add_merge:
                    m = inp[0]
                    for t in inp[1:]:
                        m = m + t
                    ans.append(m.astype(f32))
                else:
                    ans.append(np.concatenate(inp, axis=-1))
            ans_convs.append(ans)
        return ans_convs[-1][0]

    def forward(self, dilated, feats):
        if self.kind == "SymNet":  # sym_net.py:55-69
            ans = self.hrnet_forward(dilated, feats, self.layer_channels[:-1])
            if not self.use_bnds:
                ans = np.concatenate([ans, feats[dilated[0].shape[0]:]], axis=0)
            ext = f32(self.particle_radii[0]) * f32(2)
            for i, _ in enumerate(self.layer_channels[-1][-1]):
                ans = _relu(ans)
                index = self._conv_count
                self._conv_count += 1
                ans = self._cconv(index, f"model/sym_convs/{i}", (ans * self.part_scale).astype(f32), self.all_pos,
                                  self.all_pos, ext, self.window_sym, ignore=True, symmetric=True)
            return ans
        if self.kind == "HRNet":
            return self.hrnet_forward(dilated, feats, self.layer_channels)
        if self.kind == "CConv":  # cconv.py:51-69
            pos = dilated[0]
            feats = feats[:pos.shape[0]]
            ext = f32(self.particle_radii[0]) * f32(2)
            ans_convs = [feats]
            for li in range(1, len(self.layer_channels)):
                f = _relu(ans_convs[-1])
                index = self._conv_count
                self._conv_count += 1
                ans_conv = self._cconv(index, None, f, pos, pos, ext, self.window, ignore=self.ignore_query_points)
                ans_dense = self._dense(f"model/denses/{li - 1}", f)
                if ans_dense.shape[-1] == ans_convs[-1].shape[-1]:
                    ans = ans_conv + ans_dense + ans_convs[-1]
                else:
                    ans = ans_conv + ans_dense
                ans_convs.append(ans.astype(f32))
            return ans_convs[-1]
        raise NotImplementedError(self.kind)

    def postprocess(self, out, data):  # pbf_model.py:440-489
        pos, vel, acc = data[:3]
        pcnt = pos.shape[0]
        self.num_fluid_neighbors = O.reduce_subarrays_sum(np.ones(self.fluid_nns[0].shape[0], f32),
                                                          self.fluid_nns[1])[:pcnt]
        if self.cfg.get("equivar"):  # pbf_model.py:456-463 with rot = None; losses.py:337-364
            scale = self._dense("model/scale_dens", out)
            idx, rs, _ = O.fixed_radius_search(self.all_pos, self.all_pos, float(self.cfg["particle_radii"][0]))
            cnt = np.diff(rs)
            row = np.repeat(np.arange(len(cnt)), cnt)
            dx = ((self.all_pos[idx] - self.all_pos[row]) * scale[idx]).astype(f32)
            with np.errstate(invalid="ignore", divide="ignore"):
                out = (np.stack([O.reduce_subarrays_sum(np.ascontiguousarray(dx[:, k]), rs) for k in range(3)], 1)
                       / cnt[:, None].astype(f32)).astype(f32)
        if out.shape[-1] == 1:
            out = np.repeat(out, 3, axis=-1)
        elif out.shape[-1] == 2:
            out = np.concatenate([out, out[:, :1]], axis=-1)
        self.pos_correction = (self.out_scale * out[:pcnt]).astype(f32)
        pos2, vel2 = self.integrate_pos_vel(pos, vel, acc)
        dt = f32(self.timestep)
        new_pos = (pos2 + self.pos_correction).astype(f32)
        new_vel = ((new_pos - pos) / dt).astype(f32)
        return new_pos, new_vel

    def step(self, data):
        """data = [pos, vel, acc|None, feats|None, box, box_normals] (numpy float32) -> (pos', vel')."""
        self.nns_cache = {}
        self.pairs = 0
        data = [None if x is None else np.ascontiguousarray(x, dtype=f32) for x in data]
        d = self.transform(data)
        dilated, feats = self.preprocess(d)
        out = self.forward(dilated, feats)
        self.net_output = out
        res = self.postprocess(out, d)
        return self.inv_transform(res)
