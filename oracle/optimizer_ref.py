"""TEST INFRASTRUCTURE - CPU restatement of the reference's Muon / AdamW optimizer arithmetic.

Follows shared/optimizer_setup.py: ``_zeropower_via_newtonschulz5`` :43-67 (quintic Newton-Schulz,
fp32 on CPU, bf16 on an accelerator) and ``MuonAdamW.step`` :91-169 (AdamW group :103-135, Muon
group :137-167). Functional: tensors in, tensors out; ``work_dtype=torch.bfloat16`` evaluates the
iteration with bf16 tensors on the CPU, which is the arithmetic the reference runs on a GPU.
"""
import math

import torch

NS_COEFFS = (3.4445, -4.7750, 2.0315)


def newton_schulz5(grad, steps=5, work_dtype=torch.float32):
    if grad.ndim != 2:
        raise ValueError(f"Muon update expects 2D tensors, got shape {tuple(grad.shape)}")
    x = grad.to(work_dtype)
    transposed = x.size(0) > x.size(1)
    if transposed:
        x = x.transpose(0, 1)
    eps = torch.tensor(1e-7, dtype=x.dtype)
    x = x / (x.norm() + eps)
    a, b, c = NS_COEFFS
    for _ in range(max(1, int(steps))):
        xx_t = x @ x.transpose(0, 1)
        poly = b * xx_t + c * (xx_t @ xx_t)
        x = a * x + poly @ x
    if transposed:
        x = x.transpose(0, 1)
    return x.to(grad.dtype)


def adamw_update(p, grad, exp_avg, exp_avg_sq, step, lr, betas, eps, weight_decay):
    """In place on p / exp_avg / exp_avg_sq; ``step`` is the 1-based count of this update."""
    b1, b2 = betas
    exp_avg.mul_(b1).add_(grad, alpha=1.0 - b1)
    exp_avg_sq.mul_(b2).addcmul_(grad, grad, value=1.0 - b2)
    denom = (exp_avg_sq.sqrt() / math.sqrt(1.0 - b2 ** step)).add_(eps)
    if weight_decay != 0.0:
        p.mul_(1.0 - lr * weight_decay)
    p.addcdiv_(exp_avg, denom, value=-(lr / (1.0 - b1 ** step)))


def muon_update(p, grad, buf, lr, weight_decay, momentum=0.95, nesterov=True, backend_steps=5,
                work_dtype=torch.float32):
    """In place on p / buf."""
    buf.lerp_(grad, 1.0 - momentum)
    update = grad.lerp(buf, momentum) if nesterov else buf
    update = newton_schulz5(update, backend_steps, work_dtype)
    rows, cols = p.shape
    if weight_decay != 0.0:
        p.mul_(1.0 - lr * weight_decay)
    p.add_(update, alpha=-lr * max(1.0, rows / max(1, cols)))


def clip_coef(grads, max_norm):
    """Scale factor of ``clip_grad_norm_`` (trainer_backbone.py:446-449): min(1, max_norm / (norm + 1e-6))."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    return torch.clamp(max_norm / (total + 1e-6), max=1.0)
