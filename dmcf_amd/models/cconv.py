"""CConv -- mirror of the reference's ``models/cconv.py:12-69``: the single-scale residual CConv + Dense
stack (the baseline architecture of Ummenhofer et al.), fluid particles only."""
import numpy as np
import torch

from .base_model import Dense
from .pbf_model import PBFNet


class CConv(PBFNet):
    def __init__(self, name="CConv", layer_channels=[32, 64, 64, 3], window=None, out_activation=None, **kwargs):
        self.layer_channels = layer_channels
        if out_activation == "tanh":
            self.out_activation = torch.tanh
        elif out_activation is None:
            self.out_activation = lambda x: x
        else:
            raise NotImplementedError()
        torch.nn.Module.__init__(self)
        super().__init__(name=name, channels=layer_channels[0], window=window, **kwargs)

    def setup(self):
        self.convs = []
        self.denses = []
        for i in range(1, len(self.layer_channels)):  # cconv.py:36-49
            ch = self.layer_channels[i]
            self.convs.append(self.get_cconv(name="conv{0}".format(i), filters=ch, activation=None,
                                             window_func=self.window, ignore_query_points=self.ignore_query_points,
                                             circular=self.circular))
            self.denses.append(Dense(units=ch, name="dense{0}".format(i)))
        self._conv_modules = torch.nn.ModuleList([c for _, c in self._all_convs])
        self._dense_modules = torch.nn.ModuleList(self.denses)

    def forward(self, prev, data, training=True, **kwargs):
        pos, feats = prev[:2]
        pos = pos[0]
        feats = feats[:pos.shape[0]]
        filter_extent = float(np.float32(self.particle_radii[0]) * np.float32(2))
        ans_convs = [feats]
        for conv, dense in zip(self.convs, self.denses):  # cconv.py:59-67
            feats = torch.relu(ans_convs[-1])
            ans_conv = self.apply_conv(conv, feats, pos, pos, filter_extent)
            ans_dense = dense(feats)
            if ans_dense.shape[-1] == ans_convs[-1].shape[-1]:
                ans = ans_conv + ans_dense + ans_convs[-1]
            else:
                ans = ans_conv + ans_dense
            ans_convs.append(ans)
        return self.out_activation(ans_convs[-1])
