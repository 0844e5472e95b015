"""Oracle for the optimizer side of the step (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates utils/trainer_utils.py:9-49 (AdamW via HF ``Trainer.get_optimizer_cls_and_kwargs``:
betas (0.9, 0.999), eps 1e-8, decoupled weight decay, bias correction, fp32 state),
the global-norm clip of opadpo/dpo_models/rl_trainer.py:164-171 and HF's
``get_scheduler("cosine")`` lambda (transformers/optimization.py, cosine with warmup).
"""
from __future__ import annotations

import math

import torch


def cosine_lr(step: int, base_lr: float, warmup: int, total: int, num_cycles: float = 0.5) -> float:
    """LR *after* `step` scheduler steps (LambdaLR semantics: lr_0 = base*lambda(0))."""
    if step < warmup:
        return base_lr * step / max(1, warmup)
    prog = (step - warmup) / max(1, total - warmup)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * num_cycles * 2.0 * prog)))


def clip_coef(grad_sumsq: float, max_norm: float) -> float:
    """torch.nn.utils.clip_grad_norm_: coef = min(1, max_norm / (norm + 1e-6))."""
    return min(1.0, max_norm / (math.sqrt(grad_sumsq) + 1e-6))


def adamw_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float,
               beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, grad_scale=1.0):
    """One torch.optim.AdamW update on fp32 tensors (in place), `step` is 1-based."""
    g = g * grad_scale
    p.mul_(1 - lr * weight_decay)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
    return p, m, v
