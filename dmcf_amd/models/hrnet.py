"""HRNet -- mirror of the reference's ``models/hrnet.py:12-133``: the multi-resolution CConv grid.

``convs[layer][scale][k][inp_scale]``: per (layer, output scale) one CConv per input scale
(pos[inp_scale] -> pos[scale], extent 2*r[max(inp_scale, scale)], hrnet.py:85-92), a Dense + identity
residual on the same-scale branch (:93-99), merged by add_n (``add_merge``) or concat (:115-118).
"""
import os

import numpy as np
import torch

from .base_model import Dense
from .pbf_model import PBFNet


# DMCF_FUSE_EPILOGUE=0: every term of a layer's sum as its own tensor, summed by elementwise kernels (the reference's form)
_FUSE_EPILOGUE = os.environ.get("DMCF_FUSE_EPILOGUE", "1") != "0"


class HRNet(PBFNet):
    def __init__(self, name="HRNet", layer_channels=[[16], [32], [32], [3]], window=None, window_dens=None,
                 circular=False, add_merge=False, out_activation=None, **kwargs):
        self.layer_channels = layer_channels
        self.add_merge = add_merge
        if out_activation == "tanh":
            self.out_activation = torch.tanh
        elif out_activation is None:
            self.out_activation = lambda x: x
        else:
            raise NotImplementedError()
        torch.nn.Module.__init__(self)  # attributes above are plain python; modules are created in super().__init__
        super().__init__(name=name, channels=layer_channels[0][0][0], window=window, window_dens=window_dens,
                         circular=circular, **kwargs)

    def setup(self):
        """hrnet.py:39-67 (construction order == checkpoint key order of ``_all_convs``)."""
        self.convs = []
        self.denses = []
        for i in range(1, len(self.layer_channels)):
            self.denses.append([])
            self.convs.append([])
            for j in range(len(self.layer_channels[i])):
                self.convs[-1].append([])
                self.denses[-1].append([])
                for k in range(len(self.layer_channels[i][j])):
                    ch = self.layer_channels[i][j][k]
                    self.convs[-1][-1].append([])
                    self.denses[-1][-1].append([])
                    for l in range(len(self.layer_channels[i - 1]) if k == 0 else 1):
                        conv = self.get_cconv(name="conv{0}{1}{2}_{3}".format(i, j, k, l), filters=ch,
                                              activation=None, window_func=self.window,
                                              ignore_query_points=self.ignore_query_points and (j == l or k > 0),
                                              circular=self.circular)
                        self.convs[-1][-1][-1].append(conv)
                        self.denses[-1][-1][-1].append(Dense(units=ch, name="dense{0}{1}{2}_{3}".format(i, j, k, l)))
        # register for torch (parameters(), state_dict()) without changing the nested-list access pattern
        self._conv_modules = torch.nn.ModuleList([c for _, c in self._all_convs])
        self._dense_modules = torch.nn.ModuleList(
            [d for a in self.denses for b in a for c in b for d in c])

    def forward(self, prev, data, training=True, **kwargs):
        pos, feats, idx, dens = prev
        if not self.use_bnds:
            feats = feats[:pos[0].shape[0]]
        filter_extent = [float(np.float32(r) * np.float32(2)) for r in self.particle_radii]
        ans_convs = [[feats]]
        ext = None
        stash = {}       # (layer, inp_scale) -> output of a scale-0 conv of that layer computed one layer early (cross_pairs)
        relu_next = {}   # inp_scale -> relu of the next layer's input at that scale, when already formed
        for layer in range(len(self.convs)):
            n_scales = len(self.convs[layer])
            ans = [None] * n_scales
            # relu(x_{inp_scale}) is formed once per layer, not once per (scale, inp_scale) as the reference does (:85): the
            # same values, two thirds fewer elementwise kernels
            relu_in = [relu_next.pop(i) if i in relu_next else torch.relu(t) for i, t in enumerate(ans_convs[-1])]
            relu_next = {}
            if self.ghost_prefetch is not None and not (self.dens_norm and dens is not None) and self.part_scale == 1.0:
                # sharded step: every input scale of the layer is known now -- start all their ghost exchanges, the
                # convolutions pick them up one by one
                self.ghost_prefetch([(relu_in[i], pos[i], max(filter_extent[max(i, s)] for s in range(n_scales)))
                                     for i in range(len(relu_in))])
            cross = self._cross_layer_pairs(layer, dens)
            # output scales >= 1 first when a scale-0 conv of this layer is paired with one of the next layer (which reads this
            # layer's output at that scale): the sums do not depend on the order
            order = list(range(1, n_scales)) + [0] if cross else list(range(n_scales))
            for scale in order:
                importance = self.part_scale if scale == 0 else 1.0
                inp = []
                n_inp = len(ans_convs[-1])

                def feats_of(inp_scale):
                    f = relu_in[inp_scale]  # :85
                    if self.dens_norm and dens is not None and inp_scale < len(dens):  # :87-89
                        f = torch.cat([f, f / dens[inp_scale] ** 2], dim=-1)
                    return f

                # ONE running sum per output scale instead of a tensor per term and an elementwise kernel per "+" (:93-99,
                # :115-118): the Dense of the layer's own scale starts it -- with the residual as the GEMM's C operand --, every
                # convolution adds its result in its epilogue (DMCF_FLAG_ACCUMULATE), the Dense bias rides on the first of
                # them.  The same terms in a different order of additions (the parity bar of the outputs is 1e-5).
                fuse = self.add_merge and _FUSE_EPILOGUE
                acc, pending_bias = None, None
                if fuse and layer < len(self.denses) and scale < n_inp:
                    own = self.denses[layer][scale][0][scale]
                    prev_out = ans_convs[-1][scale]
                    acc = own.product(feats_of(scale), prev_out if own.units == prev_out.shape[-1] else None)
                    pending_bias = own.bias
                for inp_scale in range(n_inp):
                    feats = feats_of(inp_scale)
                    ext = filter_extent[max(inp_scale, scale)]
                    conv_in = feats if importance == 1.0 else feats * importance
                    conv = self.convs[layer][scale][0][inp_scale]
                    # a layer's rows: ~33 neighbours at the base radius, 8 x / 64 x that at the wider ones (the coarser point
                    # sets keep the particles' spacing up to scale 1) -- configuration, not list contents
                    conv.row_length_hint = 2 if max(inp_scale, scale) > 0 and self.particle_radii[max(inp_scale, scale)] > self.particle_radii[0] else 1
                    if scale == 0 and (layer, inp_scale) in stash:
                        ans_conv = stash.pop((layer, inp_scale))  # came out of the previous layer's paired launch
                    elif scale == 0 and inp_scale in cross:
                        nxt = torch.relu(ans[inp_scale])  # the next layer's input at this scale: complete (scales >= 1 first)
                        relu_next[inp_scale] = nxt
                        nxt_in = nxt if importance == 1.0 else nxt * importance
                        ans_conv, later = self._paired_convs(conv, self.convs[layer + 1][0][0][inp_scale], conv_in, nxt_in,
                                                             pos[inp_scale], pos[0], ext)
                        stash[(layer + 1, inp_scale)] = later
                    else:
                        # (the same relu(x_{inp_scale}) feeds every output scale: its widest extent is a hint for the hook)
                        widest = max(filter_extent[max(inp_scale, s)] for s in range(n_scales))
                        if acc is not None:
                            conv.accumulate_into, conv.extra_bias = acc, pending_bias
                        try:
                            ans_conv = self.apply_conv(conv, conv_in, pos[inp_scale], pos[scale], ext, widest if conv_in is feats else None)
                            if acc is not None and conv.accumulate_into is None:  # (taken: the result IS the sum)
                                pending_bias = None
                        finally:  # (a call that raised -- a ghost exchange, an allocation -- must not leave the request behind)
                            conv.accumulate_into = conv.extra_bias = None
                    if fuse:
                        if acc is None:
                            acc = ans_conv
                        elif ans_conv is not acc:
                            acc = acc.add_(ans_conv)
                        if layer < len(self.denses) and scale != inp_scale and self.voxel_size is None:  # :100-113 (FPS lists)
                            if scale > inp_scale:
                                for i in range(inp_scale, scale):
                                    feats = feats[idx[i + 1][0].long()]
                                acc = acc + self.denses[layer][scale][0][inp_scale](feats)
                            else:
                                ind = idx[scale + 1][0].long()
                                for i in range(scale + 1, inp_scale):
                                    ind = ind[idx[i + 1][0].long()]
                                acc = acc.index_add(0, ind, self.denses[layer][scale][0][inp_scale](feats))
                        continue
                    if layer < len(self.denses):
                        if scale == inp_scale:  # :93-99
                            ans_conv = ans_conv + self.denses[layer][scale][0][inp_scale](feats)
                            if ans_conv.shape[-1] == ans_convs[-1][scale].shape[-1]:
                                ans_conv = ans_conv + ans_convs[-1][scale]
                        elif self.voxel_size is None:  # :100-113: Dense across scales through the FPS index lists
                            if scale > inp_scale:
                                for i in range(inp_scale, scale):
                                    feats = feats[idx[i + 1][0].long()]
                                ans_conv = ans_conv + self.denses[layer][scale][0][inp_scale](feats)
                            else:
                                ind = idx[scale + 1][0].long()
                                for i in range(scale + 1, inp_scale):
                                    ind = ind[idx[i + 1][0].long()]
                                ans_conv = ans_conv.index_add(0, ind, self.denses[layer][scale][0][inp_scale](feats))
                    inp.append(ans_conv)
                if fuse:
                    ans[scale] = acc if pending_bias is None else acc.add_(pending_bias)
                elif self.add_merge:  # :115-118
                    merged = inp[0]
                    for t in inp[1:]:
                        merged = merged + t
                    ans[scale] = merged
                else:
                    ans[scale] = torch.cat(inp, dim=-1)
                for i in range(1, len(self.convs[layer][scale])):  # :120-131 (k > 0 sub-layers)
                    ans_conv = self.apply_conv(self.convs[layer][scale][i][0], ans[scale] * importance, pos[scale], pos[scale], ext)
                    ans_conv = ans_conv + self.denses[layer][scale][i][0](ans[scale])
                    if len(ans_convs[-1]) > scale and ans_conv.shape[-1] == ans_convs[-1][scale].shape[-1]:
                        ans_conv = ans_conv + ans_convs[-1][scale]
                    ans[scale] = ans_conv
            ans_convs.append(ans)
        return self.out_activation(ans_convs[-1][0])

    # -- two layers, one walk over a neighbour list -------------------------------------------------------------------------
    def _cross_layer_pairs(self, layer, dens):
        """Input scales s >= 1 whose conv into scale 0 of THIS layer (pos[s] -> pos[0]) can share its launch with the conv of
        the NEXT layer on the same list (hrnet.py:85-92: same point sets, same radius, same flags; only features and
        weights differ).  The next layer's input at scale s is this layer's output at scale s, which does not depend on
        this layer's scale-0 convs -- so both convs can run once that output exists: one search-list walk, one geometry
        evaluation per pair instead of two (Liquid3d: conv200_2 + conv300_2, 4 + 8 input channels on the 2.9e8-pair s2 -> s0
        list).  Restricted to what one 16-channel pass of the class-sorted kernel holds, or (wide-radius lists, plain layers) one 32-channel
        walk of the pair kernel -- Liquid3d: conv200_1 + conv300_1, 8 + 16 channels on the 3e8-pair s1 -> s0 list, 7.6 ms instead
        of 4.4 + 4.9; never on the first call (the layers
        build their weights lazily).  Nothing here depends on particle counts: in a sharded step every rank takes the same
        branch (the paired launch goes through ``conv_hook`` like any other: one ghost exchange for both feature blocks)."""
        import os
        from ..utils import convolutions as _convs
        if (os.environ.get("DMCF_FUSE_CROSS_LAYER", "1") == "0" or _convs._CACHE.depth == 0
                or layer + 1 >= len(self.convs) or not self.add_merge or self.voxel_size is None
                or (self.dens_norm and dens is not None) or len(self.convs[layer][0]) != 1 or len(self.convs[layer + 1][0]) != 1):
            return set()
        out = set()
        for s in range(1, min(len(self.convs[layer][0][0]), len(self.convs[layer + 1][0][0]), len(self.convs[layer]))):
            a, b = self.convs[layer][0][0][s], self.convs[layer + 1][0][0][s]
            if a.kernel is None or b.kernel is None or not a.kernel.is_cuda:
                continue
            wa, wb = a.window_function, b.window_function
            same = (isinstance(wa, _convs.WindowFunction) and isinstance(wb, _convs.WindowFunction) and wa.name == wb.name
                    and wa.fac == wb.fac and tuple(a.kernel.shape[:3]) == (4, 4, 4) == tuple(b.kernel.shape[:3])
                    and all(getattr(a, k) == getattr(b, k) for k in (
                        "align_corners", "coordinate_mapping", "interpolation", "normalize", "use_bias",
                        "radius_search_ignore_query_points", "radius_search_metric", "use_dense_layer_for_center"))
                    and not (a.symmetric or b.symmetric or a.circular or b.circular or a.normalize
                             or a.radius_search_ignore_query_points or a.use_dense_layer_for_center)
                    and a.activation is None and b.activation is None
                    and a.kernel.shape[3] % 4 == 0 and b.kernel.shape[3] % 4 == 0
                    and a.filters + b.filters <= 64)
            # one 16-channel pass of the class-sorted kernel -- or, up to 32 channels, one walk of the pair kernel (splat F: plain
            # poly6 layers on a wide-radius list; the plane-sorted kernel's time for 64 outputs would eat the gain)
            cin = a.kernel.shape[3] + b.kernel.shape[3]
            wide = self.particle_radii[s] > self.particle_radii[0]
            same = same and (cin <= 16 or (cin <= 32 and wide and wa.name == "poly6"))
            if same:
                out.add(s)
        return out

    def _paired_convs(self, a, b, feats_a, feats_b, inp_pos, out_pos, extent):
        """conv a on feats_a and conv b on feats_b, both inp_pos -> out_pos, as ONE convolution: features [fa | fb], the two
        kernels stacked block-diagonally, outputs [conv_a | conv_b] (every product with a zero block is an exact zero, so
        each half is the sum its own layer forms, in the order of the shared list)."""
        from .. import ops
        from ..utils import convolutions as _convs
        ca, cb, oa, ob = a.kernel.shape[3], b.kernel.shape[3], a.filters, b.filters
        feats = torch.cat([feats_a, feats_b], dim=1)
        kernel = a.kernel.new_zeros((4, 4, 4, ca + cb, oa + ob))
        kernel[..., :ca, :oa] = a.kernel
        kernel[..., ca:, oa:] = b.kernel
        bias = torch.cat([a.bias, b.bias]) if a.use_bias else None
        # (the two zero blocks of the stacked kernel are neither fetched nor multiplied by the contraction)
        mask = ops.block_diagonal_tile_mask([(0, ca, 0, oa), (ca, ca + cb, oa, oa + ob)])

        def launch(f, pi, po, ext, _):
            radius = float(np.float32(0.5) * np.float32(ext))
            nns = _convs._CACHE.search(a.fixed_radius_search, pi, po, radius, distances=False)
            index, row_splits, raw_dist = nns.raw()
            return ops.cconv_forward(kernel, po, ext, pi, f, index, row_splits, neighbors_value=raw_dist,
                                     window=a.window_function.name, window_fac=a.window_function.fac,
                                     align_corners=a.align_corners, coordinate_mapping=a.coordinate_mapping,
                                     interpolation=a.interpolation, bias=bias, n_pairs_ref=_convs.pairs_ref(nns),
                                     neighbors_row_count=getattr(nns, "row_count", None), filter_tile_mask=mask,
                                     row_length_hint=a.row_length_hint)
        out = self.apply_conv(launch, feats, inp_pos, out_pos, extent)
        a.nns = b.nns = None
        return out[:, :oa], out[:, oa:]
