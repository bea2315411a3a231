from .simulator import Simulator

__all__ = ["Simulator"]
