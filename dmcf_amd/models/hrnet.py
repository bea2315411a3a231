"""HRNet -- mirror of the reference's ``models/hrnet.py:12-133``: the multi-resolution CConv grid.

``convs[layer][scale][k][inp_scale]``: per (layer, output scale) one CConv per input scale
(pos[inp_scale] -> pos[scale], extent 2*r[max(inp_scale, scale)], hrnet.py:85-92), a Dense + identity
residual on the same-scale branch (:93-99), merged by add_n (``add_merge``) or concat (:115-118).
"""
import numpy as np
import torch

from .base_model import Dense
from .pbf_model import PBFNet


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
        for layer in range(len(self.convs)):
            ans = []
            # relu(x_{inp_scale}) is formed once per layer, not once per (scale, inp_scale) as the reference does (:85): the
            # same values, two thirds fewer elementwise kernels
            relu_in = [torch.relu(t) for t in ans_convs[-1]]
            for scale in range(len(self.convs[layer])):
                importance = self.part_scale if scale == 0 else 1.0
                inp = []
                for inp_scale in range(len(ans_convs[-1])):
                    feats = relu_in[inp_scale]  # :85
                    if self.dens_norm and dens is not None and inp_scale < len(dens):  # :87-89
                        feats = torch.cat([feats, feats / dens[inp_scale] ** 2], dim=-1)
                    ext = filter_extent[max(inp_scale, scale)]
                    conv_in = feats if importance == 1.0 else feats * importance
                    # (the same relu(x_{inp_scale}) feeds every output scale: its widest extent is a hint for the hook)
                    widest = max(filter_extent[max(inp_scale, s)] for s in range(len(self.convs[layer])))
                    ans_conv = self.apply_conv(self.convs[layer][scale][0][inp_scale], conv_in, pos[inp_scale], pos[scale], ext,
                                               widest if conv_in is feats else None)
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
                if self.add_merge:  # :115-118
                    merged = inp[0]
                    for t in inp[1:]:
                        merged = merged + t
                    ans.append(merged)
                else:
                    ans.append(torch.cat(inp, dim=-1))
                for i in range(1, len(self.convs[layer][scale])):  # :120-131 (k > 0 sub-layers)
                    ans_conv = self.apply_conv(self.convs[layer][scale][i][0], ans[-1] * importance, pos[scale], pos[scale], ext)
                    ans_conv = ans_conv + self.denses[layer][scale][i][0](ans[-1])
                    if len(ans_convs[-1]) > scale and ans_conv.shape[-1] == ans_convs[-1][scale].shape[-1]:
                        ans_conv = ans_conv + ans_convs[-1][scale]
                    ans[-1] = ans_conv
            ans_convs.append(ans)
        return self.out_activation(ans_convs[-1][0])
