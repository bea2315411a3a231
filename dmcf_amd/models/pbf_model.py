"""PBFNet -- mirror of the reference's ``models/pbf_model.py:31-517`` (inference path) on PyTorch-ROCm.

Constructor keywords are the reference's (pbf_model.py:32-74) because YAML keys are passed straight
through (run_pipeline.py:114).  Per step:

  transform    pbf_model.py:252-280   translate / scale / grav_eqvar rotation
  preprocess   pbf_model.py:303-438   semi-implicit Euler, boundary crop, features, two input CConvs +
                                      two Dense, multi-scale point sets
  forward      (subclasses)           HRNet / SymNet / CConv
  postprocess  pbf_model.py:440-489   output -> position correction, new pos / vel
  inv_transform pbf_model.py:282-301

Everything particle-sized goes through the HIP library (ContinuousConv -> dmcf_amd.ops); torch does the
[N,3] elementwise plumbing and the Dense GEMMs.
"""
import os

import numpy as np
import torch

from .. import ops
from ..utils import convolutions as _convs
from ..utils.convolutions import ContinuousConv, PointSampling
from ..utils.tools.losses import (compute_density, compute_pressure, compute_transformed_dx, get_dilated_pos,
                                  get_window_func)
from .base_model import BaseModel, Dense


def align_vector(v0, v1):
    """Rotation taking direction v0 to v1 (pbf_model.py:12-28), Rodrigues form."""
    v0n = v0 / (torch.linalg.norm(v0) + 1e-9)
    v1n = v1 / (torch.linalg.norm(v1) + 1e-9)
    v = torch.linalg.cross(v0n, v1n)
    c = torch.dot(v0n, v1n)
    s = torch.linalg.norm(v)
    if float(s) < 1e-6:
        return torch.eye(3, device=v0.device) * (-1.0 if float(c) < 0 else 1.0)
    zero = torch.zeros((), device=v0.device)
    vx = torch.stack([torch.stack([zero, -v[2], v[1]]), torch.stack([v[2], zero, -v[0]]),
                      torch.stack([-v[1], v[0], zero])])
    return torch.eye(3, device=v0.device) + vx + (vx @ vx) / (1 + c)


class PBFNet(BaseModel):
    def __init__(self, name="PBFNet", kernel_size=[4, 4, 4], channels=16, strides=[1], particle_radii=[0.05],
                 coordinate_mapping="ball_to_cube_volume_preserving", interpolation="linear", window=None,
                 window_dens=None, ignore_query_points=False, grav=-9.81, transformation={}, loss=None,
                 timestep=0.01, dens_radius=None, circular=False, dens_feats=False, pres_feats=False, equivar=False,
                 use_vel=True, use_acc=True, use_feats=False, use_box_feats=True, use_pre_adv=False, use_bnds=True,
                 dens_norm=False, rest_dens=3.5, stiffness=20.0, voxel_size=None, centralize=False,
                 out_scale=[0.01, 0.01, 0.01], sample_pad=0, sample_hyst=0.1, part_scale=1.0, **kwargs):
        super().__init__(name=name, **kwargs)
        if dens_radius is None:
            dens_radius = particle_radii
        self.equivar = equivar  # pbf_model.py:92 (False in every shipped config)
        # NN setup (pbf_model.py:79-97)
        self.kernel_size = kernel_size
        self.channel = channels
        self.strides = strides
        self.particle_radii = particle_radii
        self.coordinate_mapping = coordinate_mapping
        self.interpolation = interpolation
        self.window = window
        self.ignore_query_points = ignore_query_points
        self.voxel_size = None if voxel_size is None else np.asarray(voxel_size, dtype=np.float32)
        self.centralize = centralize
        self.circular = circular
        self.transformation = transformation or {}
        self.sample_pad = sample_pad
        self.sample_hyst = sample_hyst
        # feats setup (pbf_model.py:99-115)
        self.dens_radius = dens_radius
        self.dens_feats = dens_feats
        self.pres_feats = pres_feats
        self.dens_norm = dens_norm
        self.rest_dens = rest_dens
        self.stiffness = stiffness
        self.out_scale = [float(v) for v in out_scale]
        self.window_dens = window_dens
        self.use_vel = use_vel
        self.use_acc = use_acc
        self.use_feats = use_feats
        self.use_box_feats = use_box_feats
        self.use_pre_adv = use_pre_adv
        self.use_bnds = use_bnds
        # physics setup (pbf_model.py:117-121)
        self.timestep = timestep
        self.grav = grav
        self.part_scale = part_scale
        self.num_fluid_neighbors = 1

        self._all_convs = []  # (name, conv) in creation order == checkpoint key order (pbf_model.py:223)
        self.fluid_convs = self.get_cconv(name="fluid_obs", filters=channels, activation=None,
                                          window_func=self.window, circular=circular)
        self.fluid_dense = Dense(units=channels, name="fluid_dense")
        self.obs_convs = self.get_cconv(name="obs_conv", filters=channels, activation=None,
                                        window_func=self.window, circular=circular)
        self.obs_dense = Dense(units=channels, name="obs_dense")
        if dens_norm:  # pbf_model.py:177-181
            self.sampling = PointSampling(name="sampling", window_function=get_window_func(window_dens), normalize=True)
        if self.equivar:  # :183-189 (rot_dens is built by the reference too, its use is commented out there: :458-459)
            self.scale_dens = Dense(units=1, name="scale")
            self.rot_dens = Dense(units=4, name="rot")
        if self.use_pre_adv:  # pbf_model.py:154-175
            self.adv_convs = torch.nn.ModuleList([
                self.get_cconv(name="adv_conv0", filters=channels, activation=None, window_func=self.window,
                               circular=circular),
                self.get_cconv(name="adv_conv1", filters=channels, activation=None, window_func=self.window,
                               circular=circular)])
            self.adv_dense = torch.nn.ModuleList([Dense(units=channels, name="adv_dense0"),
                                                  Dense(units=channels, name="adv_dense1")])
        self.setup()

    def setup(self):
        return

    @property
    def num_fluid_neighbors(self):
        """Fluid neighbours per fluid particle of the last step (pbf_model.py:450-453)."""
        if self._num_fluid_neighbors is None and getattr(self, "_fluid_counts", None) is not None:
            self._num_fluid_neighbors, self._fluid_counts = self._fluid_counts(), None
        return self._num_fluid_neighbors

    @num_fluid_neighbors.setter
    def num_fluid_neighbors(self, value):
        self._num_fluid_neighbors = value
        self._fluid_counts = None

    def get_cconv(self, name, kernel_size=None, activation=None, ignore_query_points=None, window_func=None,
                  normalize=False, **kwargs):
        """pbf_model.py:197-224: the factory fixing every CConv flag of the hot path."""
        if kernel_size is None:
            kernel_size = self.kernel_size
        if ignore_query_points is None:
            ignore_query_points = self.ignore_query_points
        conv = ContinuousConv(name=name, kernel_size=kernel_size, activation=activation, align_corners=True,
                              interpolation=self.interpolation, coordinate_mapping=self.coordinate_mapping,
                              normalize=normalize, window_function=get_window_func(window_func),
                              radius_search_ignore_query_points=ignore_query_points,
                              use_dense_layer_for_center=False, **kwargs)
        self._all_convs.append((name, conv))
        return conv

    conv_hook = None
    ghost_prefetch = None  # set together with conv_hook by the sharded driver: see HRNet.forward
    # The sharded driver (dmcf_amd/parallel.py: ShardedSimulator) installs itself here for the duration of a step; preprocess asks
    # it for the three things that need other ranks -- fluid_bounds(mn, mx), begin(all_pos, pos, box), dilated_pos(base) -- and
    # every convolution goes through apply_conv / conv_hook.  None: one rank, the reference's own code path.
    shard = None

    def apply_conv(self, conv, feats, inp_pos, out_pos, extent, widest_extent=None):
        """Every ContinuousConv call of the forward pass goes through here: ``conv(feats, inp_pos, out_pos, extent, None)``
        (hrnet.py:90-92, sym_net.py:66, cconv.py) unless a ``conv_hook`` is installed -- the sharded driver
        (dmcf_amd/parallel.py) installs one that extends the input rows by the ghost particles within extent / 2.
        ``widest_extent``: the largest extent of the calls that read the same ``feats`` (a hint for that hook)."""
        if self.conv_hook is not None:
            return self.conv_hook(conv, feats, inp_pos, out_pos, extent, widest_extent)
        return conv(feats, inp_pos, out_pos, extent, None)

    def integrate_pos_vel(self, pos1, vel1, acc1=None):
        """Semi-implicit Euler (pbf_model.py:234-240)."""
        dt = self.timestep
        if acc1 is None:
            acc1 = ops.const_tensor([0.0, self.grav, 0.0], torch.float32, pos1.device)
        vel2 = vel1 + dt * acc1
        pos2 = pos1 + dt * vel2
        return pos2, vel2

    def compute_new_pos_vel(self, pos1, vel1, pos2, vel2, pos_correction):
        """pbf_model.py:242-250."""
        dt = self.timestep
        pos = pos2 + pos_correction
        vel = (pos - pos1) / dt
        return pos, vel

    def transform(self, data, training=True, **kwargs):
        pos, vel, acc, feats, box, bfeats = data
        dev = pos.device
        if "translate" in self.transformation:  # pbf_model.py:255-259
            translate = ops.const_tensor(self.transformation["translate"], torch.float32, dev)
            pos = pos + translate
            box = box + translate
        if "scale" in self.transformation:  # :261-267
            scale = ops.const_tensor(self.transformation["scale"], torch.float32, dev)
            pos = pos * scale
            box = box * scale
            vel = vel * scale
            if acc is not None:
                acc = acc * scale
        if "grav_eqvar" in self.transformation:  # :269-278
            grav_eqvar = ops.const_tensor(self.transformation["grav_eqvar"], torch.float32, dev)
            self.R = align_vector(grav_eqvar, acc[0])
            pos, vel, acc, box, bfeats = (x @ self.R for x in (pos, vel, acc, box, bfeats))
        return [pos, vel, acc, feats, box, bfeats]

    def inv_transform(self, prev, data, **kwargs):
        pos, vel = prev
        dev = pos.device
        if "grav_eqvar" in self.transformation:  # pbf_model.py:285-289
            R = self.R.t()
            pos = pos @ R
            vel = vel @ R
        if "scale" in self.transformation:  # :291-294
            scale = ops.const_tensor([max(float(v), 1e-5) for v in self.transformation["scale"]], torch.float32, dev)
            pos = pos / scale
            vel = vel / scale
        if "translate" in self.transformation:  # :296-299
            pos = pos - ops.const_tensor(self.transformation["translate"], torch.float32, dev)
        return pos, vel

    def preprocess(self, data, training=True, vel_corr=None, tape=None, **kwargs):
        _pos, _vel, acc, feats, box, bfeats = data
        if vel_corr is not None:
            vel = vel_corr
            pos = _pos + vel * self.timestep
        else:
            pos, vel = self.integrate_pos_vel(_pos, _vel, acc)  # :318
        filter_extent = [float(np.float32(r) * np.float32(2)) for r in self.particle_radii]  # :328
        # boundary particles outside the fluid AABB +- 2 r_max are dropped every step (:330-336)
        if not pos.shape[0]:  # (a rank of a sharded step may own no fluid at all)
            mn, mx = pos.new_full((3,), 3.0e38), pos.new_full((3,), -3.0e38)
        elif pos.is_cuda:
            mn, mx = ops.points_aabb(pos)  # (dmcf_points_aabb: two small launches)
        else:  # (host tensors: the CPU tests' oracle backend)
            mn, mx = torch.aminmax(pos.t().contiguous(), dim=1)
        if self.shard is not None:
            mn, mx = self.shard.fluid_bounds(mn, mx)
        lo = mn - filter_extent[-1]
        hi = mx + filter_extent[-1]
        fltr = ((box >= lo) & (box <= hi)).all(dim=1)
        keep = fltr.nonzero().flatten()  # (ONE host round trip for the count, not one per masked tensor)
        box = box.index_select(0, keep)
        bfeats = bfeats.index_select(0, keep)

        fluid_feats = [torch.ones_like(pos[:, :1])]  # :338-347
        if self.use_vel:
            fluid_feats.append(vel)
        if self.use_acc:
            if acc is None:
                raise ValueError("use_acc=True needs per-particle accelerations (pbf_model.py:341-342)")
            fluid_feats.append(acc)
        if self.use_feats:
            fluid_feats.append(feats)
        box_feats = [torch.ones_like(box[:, :1])]
        if self.use_box_feats:
            box_feats.append(bfeats)
        all_pos = torch.cat([pos, box], dim=0)  # :349
        self.all_pos = all_pos
        if self.shard is not None:
            self.shard.begin(all_pos, pos, box)
        dens = None
        if self.dens_feats or self.dens_norm or self.pres_feats:  # :351-365
            win = get_window_func(self.window_dens)
            dens = compute_density(all_pos, all_pos, self.dens_radius[0], win=win)
            n_fluid = pos.shape[0]
            if self.dens_feats:
                fluid_feats.append(dens[:n_fluid].unsqueeze(-1))
                box_feats.append(dens[n_fluid:].unsqueeze(-1))
            if self.pres_feats:
                pres = compute_pressure(all_pos, all_pos, dens, self.rest_dens, win=win, stiffness=self.stiffness)
                fluid_feats.append(pres[:n_fluid].unsqueeze(-1))
                box_feats.append(pres[n_fluid:].unsqueeze(-1))
        fluid_feats = torch.cat(fluid_feats, dim=-1)
        box_feats = torch.cat(box_feats, dim=-1)
        self.inp_feats = fluid_feats
        self.inp_bfeats = box_feats

        fused = self._fused_input_convs(fluid_feats, box_feats, pos, all_pos, filter_extent[0])
        if fused is not None:
            ans_conv, ans_obs = fused
        else:
            ans_conv = self.apply_conv(self.fluid_convs, fluid_feats * self.part_scale, pos, all_pos, filter_extent[0])  # :378
            ans_obs = self.apply_conv(self.obs_convs, box_feats * self.part_scale, box, all_pos, filter_extent[0])  # :382
        ans_dense = self.fluid_dense(fluid_feats)
        ans_dense_obs = self.obs_dense(box_feats)
        ans_dense = torch.cat([ans_dense, ans_dense_obs], dim=0)
        if self.use_pre_adv:  # :388-399
            _all_pos = torch.cat([_pos, box], dim=0)
            pre_adv_feats = torch.ones_like(_pos[:, :1])
            if self.use_vel:
                pre_adv_feats = torch.cat([pre_adv_feats, _vel], dim=-1)
            ans_adv = self.adv_convs[0](pre_adv_feats * self.part_scale, _pos, all_pos, filter_extent[0], None)
            ans_dens_adv = torch.cat([self.adv_dense[0](pre_adv_feats), ans_dense_obs], dim=0)
            fluid_feats = torch.cat([ans_conv, ans_obs, ans_adv, ans_dense, ans_dens_adv], dim=-1)
        else:
            fluid_feats = torch.cat([ans_conv, ans_obs, ans_dense], dim=-1)  # :411

        if self.shard is not None:
            dilated_pos, idx = self.shard.dilated_pos(all_pos if self.use_bnds else pos)
        else:
            dilated_pos, _, idx = get_dilated_pos(all_pos if self.use_bnds else pos, self.strides,
                                                  voxel_size=self.voxel_size, centralize=self.centralize,
                                                  pad=self.sample_pad, hyst=self.sample_hyst)  # :413-419
        if self.dens_norm:  # :421-431
            dens = [(dens if self.use_bnds else dens[:pos.shape[0]]).unsqueeze(-1)]
            for scale in range(1, len(self.dens_radius)):
                d = self.sampling(dens[-1], dilated_pos[scale - 1], dilated_pos[scale], self.dens_radius[scale], None)
                dens.append(torch.clamp(d, min=1e-2))
        else:
            dens = None
        self.dilated_pos = dilated_pos
        return [dilated_pos, fluid_feats, idx, dens]

    def fused_input_operands(self, fluid_feats, box_feats):
        """Operands of the two input layers (pbf_model.py:378-383: fluid -> all, boundary -> all, same radius, same flags)
        as ONE convolution: features [f | 0] for the fluid rows and [0 | b] for the boundary rows, the two kernels stacked
        block-diagonally, outputs [conv_f | conv_b].  Every product with a zero block is an exact zero, so each half is the
        sum the separate layer forms (in the order of the shared list).  None when the layers differ in anything but
        their weights (then they run one after the other)."""
        a, b = self.fluid_convs, self.obs_convs
        # (nothing here may depend on how many particles there are: in a sharded step every rank must take the same branch,
        # the ghost plans behind the two forms are different collectives)
        if os.environ.get("DMCF_FUSE_INPUT_CONVS", "1") == "0" or _convs._CACHE.depth == 0 or not fluid_feats.is_cuda:
            return None
        if ops.search_set() != "distance":
            # an emulation of open3d's hash walk: what a query sees depends on the TABLE of the point set searched (n / 64
            # bins), so the fluid -> all and boundary -> all lists are not the two halves of the all -> all list
            return None
        wa, wb = a.window_function, b.window_function
        if a.kernel is None or b.kernel is None:
            return None  # first call: the layers build their weights from the input widths
        same = (isinstance(wa, _convs.WindowFunction) and isinstance(wb, _convs.WindowFunction) and wa.name == wb.name
                and wa.fac == wb.fac and tuple(a.kernel.shape[:3]) == tuple(b.kernel.shape[:3]) and a.filters == b.filters
                and all(getattr(a, k) == getattr(b, k) for k in (
                    "align_corners", "coordinate_mapping", "interpolation", "normalize", "use_bias",
                    "radius_search_ignore_query_points", "radius_search_metric", "use_dense_layer_for_center"))
                and not (a.symmetric or b.symmetric or a.circular or b.circular or a.normalize
                         or a.radius_search_ignore_query_points or a.use_dense_layer_for_center)
                and a.activation is None and b.activation is None
                and a.kernel.shape[3] == fluid_feats.shape[1] and b.kernel.shape[3] == box_feats.shape[1])
        if not same:
            return None
        n, m, cf, cb, co = fluid_feats.shape[0], box_feats.shape[0], fluid_feats.shape[1], box_feats.shape[1], a.filters
        cin = -(-(cf + cb) // 4) * 4  # the matrix-core kernels take multiples of 4 channels
        feats = fluid_feats.new_zeros((n + m, cin))
        feats[:n, :cf] = fluid_feats * self.part_scale
        feats[n:, cf:cf + cb] = box_feats * self.part_scale
        kernel = a.kernel.new_zeros(tuple(a.kernel.shape[:3]) + (cin, 2 * co))
        kernel[..., :cf, :co] = a.kernel
        kernel[..., cf:cf + cb, co:] = b.kernel
        bias = torch.cat([a.bias, b.bias]) if a.use_bias else None
        return feats, kernel, bias

    def fused_input_conv(self, kernel, bias, feats, inp_pos, out_pos, extent):
        """The convolution of :meth:`fused_input_operands` on the (inp_pos -> out_pos) list of the step's cache; returns
        (output [n_out, 2 C], the list)."""
        a = self.fluid_convs
        radius = float(np.float32(0.5) * np.float32(extent))
        nns = _convs._CACHE.search(a.fixed_radius_search, inp_pos, out_pos, radius, distances=False)
        index, row_splits, raw_dist = nns.raw()
        out = ops.cconv_forward(kernel, out_pos, extent, inp_pos, feats, index, row_splits, neighbors_value=raw_dist,
                                window=a.window_function.name, window_fac=a.window_function.fac,
                                align_corners=a.align_corners, coordinate_mapping=a.coordinate_mapping,
                                interpolation=a.interpolation, bias=bias, n_pairs_ref=_convs.pairs_ref(nns),
                                neighbors_row_count=getattr(nns, "row_count", None))
        a.nns = self.obs_convs.nns = None
        return out, nns

    def _fused_input_convs(self, fluid_feats, box_feats, pos, all_pos, extent):
        """Inside a rollout step the two input layers run as one convolution on the all -> all list of the first HRNet
        layer: two searches, two grid builds and one walk over ~3e7 pairs less per step."""
        operands = self.fused_input_operands(fluid_feats, box_feats)
        if operands is None:
            return None
        feats, kernel, bias = operands
        n, co = fluid_feats.shape[0], self.fluid_convs.filters
        if self.conv_hook is not None:
            # a sharded step: through the hook like every other convolution (one ghost exchange for both feature blocks); the
            # fluid-neighbour counts below are a training-loss weight and are not formed there
            out = self.apply_conv(lambda f, pi, po, ext, _: self.fused_input_conv(kernel, bias, f, pi, po, ext)[0],
                                  feats, all_pos, all_pos, extent)
            self._fluid_counts = None
            return out[:, :co].contiguous(), out[:, co:].contiguous()
        out, nns = self.fused_input_conv(kernel, bias, feats, all_pos, all_pos, extent)
        # fluid neighbours per fluid particle (pbf_model.py:450-453; only the training loss reads them): counted from the
        # shared list when first asked for -- the closure keeps the list's index buffer alive until the next step
        index = nns.raw()[0]
        row_count = getattr(nns, "row_count", None)
        if row_count is not None:
            stride = nns.stride

            def count():
                rows = index[:n * stride].view(n, stride)
                cols = torch.arange(stride, device=index.device, dtype=torch.int32)
                return ((rows < n) & (cols[None, :] < row_count[:n, None])).sum(dim=1).to(torch.float32)
        else:
            idx, rs = nns.neighbors_index, nns.neighbors_row_splits

            def count():
                return ops.reduce_subarrays_sum((idx < n).to(torch.float32), rs)[:n]
        self._fluid_counts = count
        return out[:, :co].contiguous(), out[:, co:].contiguous()

    def postprocess(self, prev, data, training=True, vel_corr=None, **kwargs):
        pos, vel, acc = data[:3]
        pcnt = pos.shape[0]
        # number of fluid neighbours per particle (loss weight only; pbf_model.py:450-453)
        if self.shard is not None:
            self._num_fluid_neighbors = None  # (a sharded step is inference only)
        elif self.fluid_convs.nns is None and getattr(self, "_fluid_counts", None) is not None:
            self._num_fluid_neighbors = None  # formed by the property below when somebody reads it
        else:
            counts = ops.neighbor_counts(self.fluid_convs.nns)
            self.num_fluid_neighbors = counts[:pcnt]

        out = prev
        if self.equivar:  # :456-463: the network's output scales the mean offset to the neighbours (rot stays None there too)
            if self.shard is not None:
                raise NotImplementedError("equivar in a sharded step (its search has no ghost plan)")
            scale = self.scale_dens(out)
            if scale.shape[0] != self.all_pos.shape[0]:
                # use_bnds=False: the network's output has the fluid rows only while compute_transformed_dx gathers scale[idx]
                # with neighbour indices over ALL points (losses.py:310-336).  The reference has the same mismatch; TensorFlow's
                # GPU gather returns zeros for the out-of-range rows where a device gather here would read out of bounds
                raise NotImplementedError("equivar with use_bnds=False: the scale has %d rows for %d points (the reference's own "
                                          "gather is out of range there)" % (scale.shape[0], self.all_pos.shape[0]))
            out = compute_transformed_dx(self.all_pos, scale, None, radius=self.particle_radii[0])
        if out.shape[-1] == 1:  # :466-469
            out = out.repeat(1, 3)
        elif out.shape[-1] == 2:
            out = torch.cat([out, out[:, :1]], dim=-1)
        out_scale = ops.const_tensor(self.out_scale, torch.float32, pos.device)
        self.net_output = out
        self.pos_correction = out_scale * out[:pcnt]  # :474
        self.obs = out_scale * out[pcnt:]
        if vel_corr is not None:
            vel2 = vel_corr
            pos2 = pos + vel2 * self.timestep
        else:
            pos2, vel2 = self.integrate_pos_vel(pos, vel, acc)  # :484
        pos2_corrected, vel2_corrected = self.compute_new_pos_vel(pos, vel, pos2, vel2, self.pos_correction)
        return [pos2_corrected, vel2_corrected]
