#!/usr/bin/env python3
"""F15: first-stage golden vectors. Imports the reference's own `ldm.modules.diffusionmodules.model.Encoder / Decoder` from
/root/reference (read-only), fills them from the PRNG recipe under the VQModelInterface state-dict names (`encoder.*`, `decoder.*`)
and stores their outputs. `ldm.models.autoencoder` itself needs pytorch_lightning + taming (absent): the quantiser and the 1x1 glue
convs are not pinned here (oracle/vq.py says so).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_vq.py
"""
from __future__ import annotations

import contextlib
import io
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
from tests.golden.summary import summarize  # noqa: E402

torch.set_grad_enabled(False)
CASES = {  # tag: (ddconfig, batch, image side)
    "tiny": (dict(double_z=False, z_channels=3, resolution=64, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 4], num_res_blocks=1,
                  attn_resolutions=[], dropout=0.0), 2, 64),
    # the shipped architecture (vq-f4.yaml ddconfig) at a 128^2 image (32^2 latent) instead of 512^2
    "f4": (dict(double_z=False, z_channels=3, resolution=128, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4], num_res_blocks=2,
                attn_resolutions=[], dropout=0.0), 1, 128),
}


def fill(module, prefix, seed):
    for name, p in module.named_parameters():
        p.copy_(prng.fill_value(seed, prefix + name, p.shape))


def main():
    from ldm.modules.diffusionmodules import model as rm
    for tag, (dd, B, side) in CASES.items():
        with contextlib.redirect_stdout(io.StringIO()):
            enc, dec = rm.Encoder(**dd).eval(), rm.Decoder(**dd).eval()
        fill(enc, "encoder.", 15); fill(dec, "decoder.", 15)
        x = prng.uniform(15, f"vq.{tag}.x", (B, 3, side, side))
        z = prng.normal(15, f"vq.{tag}.z", (B, 3, side // 4, side // 4))
        ze, yd = enc(x), dec(z)
        out = {"enc_out": ze.numpy(), "n_enc": sum(p.numel() for p in enc.parameters()), "n_dec": sum(p.numel() for p in dec.parameters())}
        if yd.numel() <= 40000:
            out["dec_out"] = yd.numpy()
        for k, v in summarize(yd).items():
            out[f"dec_out.{k}"] = v
        np.savez(os.path.join(HERE, f"f15_vq_{tag}.npz"), **{k: np.asarray(v) for k, v in out.items()})
        print(f"wrote f15_vq_{tag}.npz", os.path.getsize(os.path.join(HERE, f"f15_vq_{tag}.npz")), "bytes")


if __name__ == "__main__":
    main()
