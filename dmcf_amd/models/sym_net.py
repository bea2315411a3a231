"""SymNet -- mirror of the reference's ``models/sym_net.py:12-69``: HRNet trunk + antisymmetric CConv
(ASCC) head that conserves momentum (sum_i dx_i = 0 over fluid + boundary particles)."""
import numpy as np
import torch

from .hrnet import HRNet


class SymNet(HRNet):
    def __init__(self, name="SymNet", layer_channels=[[[16]], [[32]], [[32]], [[3]]], sym_kernel_size=[6, 6, 6],
                 sym_axis=2, window_sym=None, out_activation=None, **kwargs):
        self.sym_kernel_size = sym_kernel_size
        self.sym_axis = sym_axis
        self.window_sym = window_sym
        self.sym_channels = layer_channels[-1][-1]
        if out_activation == "tanh":
            self.act = torch.tanh
        elif out_activation is None:
            self.act = lambda x: x
        else:
            raise NotImplementedError()
        super().__init__(name=name, layer_channels=layer_channels[:-1], out_activation=None, **kwargs)

    def setup(self):
        super().setup()
        self.sym_convs = []
        for i, ch in enumerate(self.sym_channels):  # sym_net.py:42-53
            conv = self.get_cconv(name="sym_conv{0}".format(i), filters=ch, activation=None, use_bias=False,
                                  symmetric=True, kernel_size=self.sym_kernel_size, ignore_query_points=True,
                                  window_func=self.window_sym, sym_axis=self.sym_axis, circular=self.circular)
            self.sym_convs.append(conv)
        self._conv_modules = torch.nn.ModuleList([c for _, c in self._all_convs])

    def forward(self, prev, data, training=True, **kwargs):
        pos, feats, idx, dens = prev
        ans = super().forward(prev, data, training, **kwargs)
        if not self.use_bnds:
            ans = torch.cat([ans, feats[pos[0].shape[0]:]], dim=0)
        ext = float(np.float32(self.particle_radii[0]) * np.float32(2))
        for conv in self.sym_convs:  # sym_net.py:63-67
            ans = torch.relu(ans)
            conv_in = ans if self.part_scale == 1.0 else ans * self.part_scale
            ans = self.apply_conv(conv, conv_in, self.all_pos, self.all_pos, ext)
        return self.act(ans)
