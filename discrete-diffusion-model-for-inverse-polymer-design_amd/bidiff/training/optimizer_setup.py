"""Optimizer construction for backbone training: drop-in for ``build_step1_backbone_optimizer``
(shared/optimizer_setup.py:280-317). Plain AdamW (default) gives ``FlatAdamW``; the ``muon_adamw``
type splits the parameters as the reference does (:172-277: 2-D matrices inside the transformer
blocks go to Muon, embeddings / output projection / biases / LayerNorms to AdamW) and gives
``FlatMuonAdamW``. Returns ``(optimizer, info)`` with the reference's info keys."""
from typing import Any, Dict, Tuple

import torch

from .optimizer import FlatAdamW, FlatMuonAdamW

_EXCLUDE = ["token_embedding", "position_embedding", "timestep_embedding", "time_embedding", "output_proj",
            "node_head", "edge_head", "edge_embedding"]


def _num(value, name, kind=float):
    try:
        return kind(value)
    except (TypeError, ValueError):
        what = "numeric" if kind is float else "integer-like"
        raise ValueError(f"{name} must be {what}, got {value!r} ({type(value).__name__})")


def _cfg(cfg, key, default):
    value = cfg.get(key, default)          # an explicit null still means the default
    return default if value is None else value


def _unwrap(model):
    m = model
    if isinstance(m, torch.nn.parallel.DistributedDataParallel):
        m = m.module
    return getattr(m, "_orig_mod", m)


def build_muon_param_groups(model, weight_decay, base_learning_rate, muon_cfg):
    layer_tokens = tuple(muon_cfg.get("layer_name_tokens", ["layers.", "blocks."]))
    exclude = tuple(muon_cfg.get("exclude_name_tokens", _EXCLUDE))
    adamw, muon = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if p.ndim == 2 and any(t in name for t in layer_tokens) and not any(t in name for t in exclude):
            muon.append(p)
        else:
            adamw.append(p)
    if not muon:
        return [], {"using_muon": False, "reason": "no_muon_eligible_parameters", "num_muon_tensors": 0,
                    "num_adamw_tensors": len(adamw)}
    adamw_lr = _num(_cfg(muon_cfg, "adamw_lr", base_learning_rate), "optimizer.muon_adamw.adamw_lr")
    if muon_cfg.get("muon_lr", None) is None:
        mult = _num(_cfg(muon_cfg, "muon_lr_multiplier", 20.0), "optimizer.muon_adamw.muon_lr_multiplier")
        muon_lr = adamw_lr * mult
    else:
        muon_lr = _num(muon_cfg["muon_lr"], "optimizer.muon_adamw.muon_lr")
    betas = _cfg(muon_cfg, "adamw_betas", [0.9, 0.95])
    if not isinstance(betas, (list, tuple)) or len(betas) != 2:
        raise ValueError("optimizer.muon_adamw.adamw_betas must be a length-2 list.")
    steps = _num(_cfg(muon_cfg, "backend_steps", 5), "optimizer.muon_adamw.backend_steps", int)
    nesterov = bool(_cfg(muon_cfg, "muon_nesterov", True))
    groups = [
        {"params": adamw, "kind": "adamw", "lr": adamw_lr, "weight_decay": weight_decay,
         "betas": (float(betas[0]), float(betas[1])),
         "eps": _num(_cfg(muon_cfg, "adamw_eps", 1.0e-8), "optimizer.muon_adamw.adamw_eps")},
        {"params": muon, "kind": "muon", "lr": muon_lr, "weight_decay": weight_decay,
         "momentum": _num(_cfg(muon_cfg, "muon_momentum", 0.95), "optimizer.muon_adamw.muon_momentum"),
         "nesterov": nesterov, "backend_steps": steps},
    ]
    return groups, {"using_muon": True, "reason": None, "num_muon_tensors": len(muon), "num_adamw_tensors": len(adamw),
                    "adamw_lr": adamw_lr, "muon_lr": muon_lr, "muon_lr_multiplier": muon_lr / max(1e-12, adamw_lr),
                    "backend_steps": steps, "muon_nesterov": nesterov}


def build_step1_backbone_optimizer(model, optimization_config: Dict[str, Any], base_learning_rate: float,
                                   weight_decay: float) -> Tuple[torch.optim.Optimizer, Dict[str, Any]]:
    ocfg = optimization_config.get("optimizer", {})
    if str(ocfg.get("type", "adamw")).strip().lower() != "muon_adamw":
        return (FlatAdamW(model.parameters(), lr=base_learning_rate, weight_decay=weight_decay),
                {"type": "adamw", "using_muon": False})
    mcfg = ocfg.get("muon_adamw", {})
    if not bool(mcfg.get("enabled", True)):
        return (FlatAdamW(model.parameters(), lr=base_learning_rate, weight_decay=weight_decay),
                {"type": "adamw", "using_muon": False, "reason": "muon_disabled"})
    groups, info = build_muon_param_groups(_unwrap(model), weight_decay, base_learning_rate, mcfg)
    if not info.get("using_muon", False):
        return (FlatAdamW(model.parameters(), lr=base_learning_rate, weight_decay=weight_decay),
                {"type": "adamw", "using_muon": False, "reason": info.get("reason")})
    return FlatMuonAdamW(groups), {"type": "muon_adamw", **info}
