"""BaseModel -- mirror of the reference's ``models/base_model.py:10-107``: the five-stage ``call``."""
import torch

from ..utils.convolutions import PlainAttributes

from .. import ops
from ..utils.config import Config


class Dense(PlainAttributes, torch.nn.Module):
    """tf.keras.layers.Dense(units, activation=None): lazily built ``kernel`` [in, units] (glorot
    uniform) and ``bias`` [units] (zeros); dmcf_dense_forward for the shapes it takes (a row per thread), else torch's GEMM."""

    def __init__(self, units, name=None, activation=None, use_bias=True):
        super().__init__()
        assert activation is None
        self.units = units
        self.layer_name = name
        self.use_bias = use_bias
        self.kernel = None
        self.bias = None

    def build(self, in_features, device):
        limit = (6.0 / (in_features + self.units)) ** 0.5
        self.kernel = torch.nn.Parameter(torch.empty(in_features, self.units, device=device).uniform_(-limit, limit),
                                         requires_grad=False)
        if self.use_bias:
            self.bias = torch.nn.Parameter(torch.zeros(self.units, device=device), requires_grad=False)

    @torch.no_grad()
    def forward(self, x):
        if self.kernel is None:
            self.build(x.shape[-1], x.device)
        if ops.dense_supported(x, self.kernel):
            return ops.dense_forward(x, self.kernel, self.bias)
        if self.bias is not None:
            return torch.addmm(self.bias, x, self.kernel)
        return x @ self.kernel

    @torch.no_grad()
    def product(self, x, residual=None):
        """``x @ kernel`` (+ ``residual``, inside the GEMM: its C operand) WITHOUT the bias -- the start of a layer's sum
        (models/hrnet.py); the bias rides on the first convolution that accumulates into the result."""
        if self.kernel is None:
            self.build(x.shape[-1], x.device)
        if ops.dense_supported(x, self.kernel):
            return ops.dense_forward(x, self.kernel, None, residual)
        if residual is not None:
            return torch.addmm(residual, x, self.kernel)
        return x @ self.kernel


class BaseModel(PlainAttributes, torch.nn.Module):
    """models/base_model.py:10-29.  ``model(data, training=False)`` runs
    transform -> preprocess -> forward -> postprocess -> inv_transform."""

    def __init__(self, name, **kwargs):
        super().__init__()
        self.model_name = name
        self.cfg = Config(kwargs)  # unknown kwargs (ckpt_path, device, ...) end up here, base_model.py:19-21

    @property
    def name(self):
        return self.model_name

    def __call__(self, data, training=True, **kwargs):
        return self.call(data, training=training, **kwargs)

    def call(self, data, training=True, **kwargs):
        d = self.transform(data, training=training, **kwargs)
        x = self.preprocess(d, training=training, **kwargs)
        x = self.run_forward(x, d, training=training, **kwargs)
        x = self.postprocess(x, d, training=training, **kwargs)
        x = self.inv_transform(x, data, training=training, **kwargs)
        return x

    # torch.nn.Module reserves ``forward`` for __call__ dispatch; the reference's stage is named
    # ``forward(prev, data, training)`` (base_model.py:31-33) and subclasses here keep that name.
    def run_forward(self, prev, data, training=True, **kwargs):
        return self.forward(prev, data, training=training, **kwargs)

    def forward(self, prev, data, training=True, **kwargs):
        raise NotImplementedError

    def loss(self, results, data):
        raise NotImplementedError("training is out of scope of the MI355X hot path (SURVEY.md section 2 row 16)")

    def get_optimizer(self, cfg_pipeline):
        raise NotImplementedError("training is out of scope of the MI355X hot path (SURVEY.md section 2 row 16)")

    def transform(self, data, training=True, **kwargs):
        return data

    def inv_transform(self, prev, data, training=True, **kwargs):
        return prev

    def preprocess(self, data, training=True, **kwargs):
        return data

    def postprocess(self, prev, data, training=True, **kwargs):
        return prev
