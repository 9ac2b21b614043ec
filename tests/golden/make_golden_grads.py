#!/usr/bin/env python3
"""F14 — gradient fixtures of the training step: the reference's own `UNetModel` is imported from
/root/reference, filled from the PRNG recipe, run forward + L1 loss + backward on the CPU, and the gradients are stored as
compact summaries (per-parameter L2 norm + 8 values at fixed flat indices; dL/dx and dL/dcontext in full or summarised).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_grads.py
Only numbers are written (f14_grads_<tag>.npz); nothing of the reference travels.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = os.environ.get("STEDM_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

from stedm_amd.utils import prng  # noqa: E402
from tests.golden.summary import summarize  # noqa: E402

N_PICK = 8


def pick_index(numel: int) -> np.ndarray:
    """8 flat indices per parameter, fixed by its size alone."""
    return np.unique(np.linspace(0, numel - 1, min(N_PICK, numel)).astype(np.int64))


def grad_summary(named_grads) -> dict:
    d = {}
    for name, g in named_grads:
        a = g.detach().double().reshape(-1).numpy()
        d[f"g.{name}.norm"] = np.float64(np.sqrt((a * a).sum()))
        d[f"g.{name}.pick"] = a[pick_index(a.size)].astype(np.float32)
    return d


def main():
    from ldm.modules.diffusionmodules import openaimodel as rom

    def run(tag, B, hw, seed, **kw):
        m = rom.UNetModel(**kw).train()      # dropout = 0: train() == eval() arithmetically; checkpointing is active as in training
        prng.fill_module_(m, seed=seed)
        x = prng.normal(seed, f"unet.{tag}.x", (B, kw["in_channels"], hw, hw)).requires_grad_(True)
        ctx = prng.normal(seed, f"unet.{tag}.ctx", (B, kw["model_channels"] * 4)).requires_grad_(True)
        target = prng.normal(seed, f"unet.{tag}.target", (B, kw["out_channels"], hw, hw))
        tt = torch.tensor(([951, 21, 500, 1] * B)[:B], dtype=torch.long)
        y = m(x, tt, context=ctx)
        # get_loss 'l1', mean=False then mean over (1,2,3), then .mean() over the batch (ddpm.py:282-295, 1030-1040; logvar = 0,
        # l_simple_weight = 1, original_elbo_weight = 0)
        loss = (y - target).abs().mean(dim=(1, 2, 3)).mean()
        loss.backward()
        d = {"t": tt.numpy(), "loss": np.float64(loss.item())}
        for k, v in summarize(x.grad).items():
            d[f"dx.{k}"] = v
        d["dctx"] = ctx.grad.numpy()
        if x.grad.numel() <= 16384:
            d["dx"] = x.grad.numpy()
        missing = [n for n, p in m.named_parameters() if p.grad is None]
        assert not missing, missing
        d.update(grad_summary((n, p.grad) for n, p in m.named_parameters()))
        path = os.path.join(HERE, f"f14_grads_{tag}.npz")
        np.savez(path, **d)
        print(f"wrote f14_grads_{tag}.npz {os.path.getsize(path) / 1024:.1f} KB  loss {loss.item():.6f}")

    tiny = dict(image_size=16, in_channels=7, model_channels=32, out_channels=4, num_res_blocks=2,
                attention_resolutions=[32, 16, 8], channel_mult=[1, 2, 4], num_heads=4)
    run("tiny", 2, 16, 6, **tiny)
    ns = dict(image_size=32, in_channels=7, model_channels=128, out_channels=4, num_res_blocks=2,
              attention_resolutions=[32, 16, 8], channel_mult=[1, 4, 8], num_heads=8)
    run("ns32", 2, 32, 0, **ns)


if __name__ == "__main__":
    main()
