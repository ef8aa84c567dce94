from .trainer_backbone import BackboneTrainer
from .optimizer import FlatAdamW

__all__ = ["BackboneTrainer", "FlatAdamW"]
