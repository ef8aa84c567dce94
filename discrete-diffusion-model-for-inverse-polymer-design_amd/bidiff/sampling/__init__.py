from .sampler import ConstrainedSampler

__all__ = ["ConstrainedSampler"]
