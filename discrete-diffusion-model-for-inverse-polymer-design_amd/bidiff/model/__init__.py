from .backbone import DiffusionBackbone
from .diffusion import DiscreteMaskingDiffusion
from .property_head import PropertyHead, PropertyPredictor

__all__ = ["DiffusionBackbone", "DiscreteMaskingDiffusion", "PropertyHead", "PropertyPredictor"]
