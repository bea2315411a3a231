"""Model classes with the reference's names (``getattr(models, cfg.model.name)``, run_pipeline.py:110)."""
from .base_model import BaseModel, Dense
from .pbf_model import PBFNet
from .hrnet import HRNet
from .sym_net import SymNet
from .cconv import CConv

__all__ = ["BaseModel", "Dense", "PBFNet", "HRNet", "SymNet", "CConv"]
