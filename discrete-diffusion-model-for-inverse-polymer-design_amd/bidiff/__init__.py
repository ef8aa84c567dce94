"""MI355X-native Bi_Diffusion hot path: denoiser forward, q(x_t|x_0) corruption, reverse
sampling step and training step, behind the reference's Python class API."""
from .model.backbone import DiffusionBackbone
from .model.diffusion import DiscreteMaskingDiffusion

__all__ = ["DiffusionBackbone", "DiscreteMaskingDiffusion"]
