#!/usr/bin/env python3
"""F13: optimizer-side golden vectors. Imports the reference's own `ldm.modules.ema.LitEma` from /root/reference (read-only) and
`torch.optim.AdamW` (the optimizer the reference constructs, modules/ldm_diffusion.py:224-234), runs both on a small parameter set
filled from the PRNG recipe and stores the traces. Only numbers are written; nothing of the reference travels.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_opt.py
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
sys.path.insert(0, os.environ.get("STEDM_REFERENCE", "/root/reference"))

from stedm_amd.utils import prng  # noqa: E402

SHAPES = {"conv.weight": (6, 4, 3, 3), "conv.bias": (6,), "lin.weight": (5, 7)}
LR, BETAS, EPS, WD = 1e-3, (0.9, 0.999), 1e-8, 1e-2      # torch.optim.AdamW defaults except lr (the reference passes lr only)
NSTEP = 3


class Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = torch.nn.Conv2d(4, 6, 3)
        self.lin = torch.nn.Linear(7, 5, bias=False)


def grads(step):
    return {n: prng.normal(13, f"f13.g{step}.{n}", s) * 0.3 for n, s in SHAPES.items()}


def main():
    from ldm.modules.ema import LitEma
    out = {}
    m = Tiny()
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(prng.normal(13, f"f13.p.{n}", SHAPES[n]))
            out[f"p0.{n}"] = p.detach().numpy().copy()
    ema = LitEma(m)                                   # decay 0.9999, use_num_upates=True (ddpm.py:86-88 builds it this way)
    opt = torch.optim.AdamW(m.parameters(), lr=LR)
    names = [n for n, _ in m.named_parameters()]
    for step in range(1, NSTEP + 1):
        g = grads(step)
        for n, p in m.named_parameters():
            p.grad = g[n].clone()
        opt.step()
        ema(m)                                        # on_train_batch_end (ddpm.py:369-371)
        sh = dict(ema.named_buffers())
        for n, p in m.named_parameters():
            out[f"p{step}.{n}"] = p.detach().numpy().copy()
            out[f"ema{step}.{n}"] = sh[ema.m_name2s_name[n]].numpy().copy()
            st = opt.state[p]
            out[f"m{step}.{n}"] = st["exp_avg"].numpy().copy()
            out[f"v{step}.{n}"] = st["exp_avg_sq"].numpy().copy()
        out[f"num_updates{step}"] = int(ema.num_updates)
    # the saturated branch of LitEma's decay (decay = min(0.9999, (1+n)/(10+n)) hits 0.9999 beyond n = 89 990)
    ema.num_updates.fill_(200000)
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.add_(prng.normal(13, f"f13.delta.{n}", SHAPES[n]))
            out[f"pL.{n}"] = p.detach().numpy().copy()
    ema(m)
    sh = dict(ema.named_buffers())
    for n in names:
        out[f"emaL.{n}"] = sh[ema.m_name2s_name[n]].numpy().copy()
    out["num_updatesL"] = int(ema.num_updates)
    out["hyper"] = np.array([LR, BETAS[0], BETAS[1], EPS, WD, 0.9999], dtype=np.float64)
    np.savez(os.path.join(HERE, "f13_ema_adamw.npz"), **out)
    print("wrote f13_ema_adamw.npz", os.path.getsize(os.path.join(HERE, "f13_ema_adamw.npz")), "bytes")


if __name__ == "__main__":
    main()
