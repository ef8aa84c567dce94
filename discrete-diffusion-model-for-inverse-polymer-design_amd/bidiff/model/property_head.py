"""Property regression on top of the denoiser: the step right after sampling in inverse design.

Drop-in for the reference's ``src/model/property_head.py``: ``PropertyHead`` (:8-69) is a small MLP
on the pooled hidden state, ``PropertyPredictor`` (:72-237) runs the backbone at a fixed timestep,
pools and applies the head. The backbone forward and the pooling are HIP kernels
(``DiffusionBackbone.get_pooled_output``); the head is a few hundred kFLOP per sequence and stays
a torch module that trains with ordinary autograd, on frozen features or - ``finetune_last_layers`` -
together with the last blocks, whose gradients the training-step backward kernels provide.
"""
from typing import Dict, List, Optional

import torch
import torch.nn as nn


class PropertyHead(nn.Module):
    """Linear -> ReLU -> Dropout per hidden size, then Linear(., 1); Xavier-uniform weights and zero
    biases in module order, so a torch seed reproduces the reference's head (property_head.py:32-58)."""

    def __init__(self, input_size: int, hidden_sizes: List[int] = [256, 64], dropout: float = 0.1):
        super().__init__()
        self.input_size = input_size
        self.hidden_sizes = hidden_sizes
        layers = []
        prev = input_size
        for h in hidden_sizes:
            layers += [nn.Linear(prev, h), nn.ReLU(), nn.Dropout(dropout)]
            prev = h
        layers.append(nn.Linear(prev, 1))
        self.mlp = nn.Sequential(*layers)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        """[batch, hidden] -> [batch, 1]."""
        return self.mlp(hidden_states)


class PropertyPredictor(nn.Module):
    """Backbone (frozen feature extractor) + pooling + head (reference property_head.py:72-194)."""

    def __init__(self, backbone: nn.Module, property_head: PropertyHead, freeze_backbone: bool = True,
                 finetune_last_layers: int = 0, pooling: str = "mean", default_timestep: int = 1):
        super().__init__()
        self.backbone = backbone
        self.property_head = property_head
        self.pooling = pooling
        self.default_timestep = default_timestep
        if freeze_backbone:
            for p in self.backbone.parameters():
                p.requires_grad = False
            if finetune_last_layers > 0:
                # the last blocks and the final norm train with the head (property_head.py:106-113): their
                # gradients come from the training-step backward kernels, which stop below the lowest
                # unfrozen block
                for layer in self.backbone.layers[-finetune_last_layers:]:
                    for p in layer.parameters():
                        p.requires_grad = True
                for p in self.backbone.final_norm.parameters():
                    p.requires_grad = True

    def forward(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                timesteps: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Predictions [batch, 1]; ``timesteps`` defaults to ``default_timestep`` for every row."""
        if timesteps is None:
            timesteps = torch.full((input_ids.shape[0],), self.default_timestep, device=input_ids.device,
                                   dtype=torch.long)
        pooled = self.backbone.get_pooled_output(input_ids, timesteps, attention_mask, self.pooling)
        return self.property_head(pooled)

    def compute_loss(self, input_ids: torch.Tensor, labels: torch.Tensor,
                     attention_mask: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """MSE against ``labels`` ([batch] or [batch, 1]); returns ``loss`` and ``predictions``."""
        predictions = self.forward(input_ids, attention_mask)
        if labels.dim() == 1:
            labels = labels.unsqueeze(-1)
        return {"loss": nn.MSELoss()(predictions, labels), "predictions": predictions}

    def predict(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Inference-mode predictions [batch] (puts the module in eval mode, as the reference does)."""
        self.eval()
        with torch.no_grad():
            predictions = self.forward(input_ids, attention_mask)
        return predictions.squeeze(-1)

    def save_pretrained(self, path: str, include_backbone: bool = False):
        """The reference's checkpoint layout (property_head.py:196-214)."""
        ckpt = {
            "property_head_state_dict": self.property_head.state_dict(),
            "pooling": self.pooling,
            "default_timestep": self.default_timestep,
            "input_size": self.property_head.input_size,
            "hidden_sizes": self.property_head.hidden_sizes,
        }
        if include_backbone:
            ckpt["backbone_state_dict"] = self.backbone.state_dict()
        torch.save(ckpt, path)

    def load_property_head(self, path: str):
        """Loads either checkpoint layout the reference accepts (property_head.py:216-237); the file is
        read with ``weights_only=True``."""
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        if "property_head_state_dict" in ckpt:
            sd = ckpt["property_head_state_dict"]
        elif "model_state_dict" in ckpt:
            ms = ckpt["model_state_dict"]
            if any(k.startswith("_orig_mod.") for k in ms):
                ms = {k.replace("_orig_mod.", ""): v for k, v in ms.items()}
            prefix = "property_head."
            sd = {k[len(prefix):]: v for k, v in ms.items() if k.startswith(prefix)}
            if not sd:
                raise KeyError("No property_head weights found in model_state_dict")
        else:
            raise KeyError("Checkpoint missing property_head_state_dict or model_state_dict")
        self.property_head.load_state_dict(sd)
