"""TEST INFRASTRUCTURE ONLY — CPU restatement of the training step's arithmetic: U-Net forward, L1 loss of
ddpm.py:1030-1040 (loss_type l1, logvar == 0, l_simple_weight 1, original_elbo_weight 0), reverse-mode gradients by autograd
over oracle/unet.py's functional forward, AdamW as torch.optim.AdamW computes it, and LitEma's update (ema.py:25-44).
Pinned by tests/golden/f14_grads_*.npz (the reference's own UNetModel run forward + backward, make_golden_grads.py) and, for the
optimizer side, by tests/golden/f13_ema_adamw.npz (traces of torch.optim.AdamW and of the reference's own LitEma, make_golden_opt.py).
Only tests/, smoke() and bench.py's cpu_baseline leg may import this package."""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from . import unet as ounet


def l1_loss(model_output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """ddpm.py:282-295 + 1030-1040: |target - out| mean over (1,2,3), then mean over the batch."""
    return (target - model_output).abs().mean(dim=[1, 2, 3]).mean()


def unet_loss_and_grads(P: Dict[str, torch.Tensor], cfg: ounet.UNetConfig, x: torch.Tensor, t: torch.Tensor, ctx: torch.Tensor,
                        target: torch.Tensor) -> Tuple[float, Dict[str, torch.Tensor], torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (loss, {param name: grad}, dL/dx, dL/dcontext, model output)"""
    Pg = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    xg = x.detach().clone().requires_grad_(True)
    cg = ctx.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        y = ounet.unet_forward.__wrapped__(Pg, cfg, xg, t, cg)
        loss = l1_loss(y, target)
        loss.backward()
    return float(loss.detach()), {k: v.grad for k, v in Pg.items()}, xg.grad, cg.grad, y.detach()


def adamw_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float, beta1: float = 0.9,
               beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 1e-2) -> None:
    """torch.optim.AdamW (the reference's optimizer, ldm_diffusion.py:224-234), single tensor, in place; `step` counts from 1."""
    p.mul_(1.0 - lr * weight_decay)
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def ema_decay(num_updates: int, decay: float = 0.9999) -> float:
    """ema.py:28-31: num_updates is the value AFTER the increment."""
    return min(decay, (1 + num_updates) / (10 + num_updates))


def ema_update(shadow: torch.Tensor, p: torch.Tensor, decay: float) -> None:
    """ema.py:40: shadow -= (1 - decay) * (shadow - p)."""
    shadow.sub_((1.0 - decay) * (shadow - p))
