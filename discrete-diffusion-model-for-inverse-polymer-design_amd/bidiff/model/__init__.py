from .backbone import DiffusionBackbone
from .diffusion import DiscreteMaskingDiffusion

__all__ = ["DiffusionBackbone", "DiscreteMaskingDiffusion"]
