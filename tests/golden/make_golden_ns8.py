#!/usr/bin/env python3
"""F7b: the set-ViT with 8 style images (BASELINE config 5: "8 style inputs set-agg"). Imports the reference's own `networks.vit_set.sViT`
from /root/reference (read-only), PRNG-recipe weights, stores the 512-d outputs: a 64^2 case and the full 512^2 / 4098-token case.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ns8.py
"""
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

torch.set_grad_enabled(False)


def main():
    from networks.vit_set import sViT
    out = {}
    for tag, (img, B) in {"i64_ns8": (64, 2), "i512_ns8": (512, 1)}.items():
        m = sViT(image_size=img, patch_size=8, num_classes=512, dim=256, depth=6, heads=12, mlp_dim=256, pool="mean",
                 channels=3, dropout=0.1, emb_dropout=0.1, ns=8, t_dim=256).eval()
        prng.fill_module_(m, seed=7)
        for l, (attn, _ff) in enumerate(m.transformer.layers):
            attn.fn.temperature.fill_(float(np.log(64 ** -0.5)) + 0.05 * l)
        x = prng.uniform(7, f"svit.{tag}.img", (B, 8, img, img, 3))
        out[tag] = m(x).numpy()
    np.savez(os.path.join(HERE, "f7_svit_ns8.npz"), **out)
    print("wrote f7_svit_ns8.npz", os.path.getsize(os.path.join(HERE, "f7_svit_ns8.npz")), "bytes")


if __name__ == "__main__":
    main()
