#!/usr/bin/env python3
"""F16: the reference's set-ViT in TRAIN mode (the reference runs the agg block inside the training step: networks/s_zss_dm.py:45-60) with the
build's dropout masks injected. Imports the reference's own `networks.vit_set.sViT` from /root/reference (read-only), PRNG-recipe weights,
`.train()`; torch.nn.functional.dropout — what every nn.Dropout of vit_set.py (:28-30, :43, :49, :187) calls — is replaced for the duration by
a function that keeps torch's arithmetic (input * mask / (1 - p)) but takes the mask from oracle/dropmask.py, the site id following the call
order of sViT.forward: emb (vit_set.py:187), then per layer attention probabilities (:62), to_out (:49), FeedForward hidden and output (:28-30).
This pins WHERE and HOW the oracle's train-mode path applies dropout to the reference module itself.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_train_drop.py
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

from oracle import dropmask  # noqa: E402
from stedm_amd.utils import prng  # noqa: E402

torch.set_grad_enabled(False)
SEED = 0x5EED00D5
SITE_EMB, SITE_ATTN, SITE_OUT, SITE_FF1, SITE_FF2 = 0x10000, 1, 2, 3, 4


def main():
    import torch.nn.functional as F
    from networks.vit_set import sViT
    out = {"seed": np.array([SEED], dtype=np.int64)}
    for tag, (img, ns, B, depth, heads) in {"i64_ns4_d2": (64, 4, 2, 2, 12), "i32_ns1_d3": (32, 1, 3, 3, 4)}.items():
        m = sViT(image_size=img, patch_size=8, num_classes=512, dim=256, depth=depth, heads=heads, mlp_dim=256, pool="mean",
                 channels=3, dropout=0.1, emb_dropout=0.1, ns=ns, t_dim=256).train()
        prng.fill_module_(m, seed=7)
        for l, (attn, _ff) in enumerate(m.transformer.layers):
            attn.fn.temperature.fill_(float(np.log(64 ** -0.5)) + 0.05 * l)
        x = prng.uniform(7, f"svit.train.{tag}.img", (B, ns, img, img, 3))
        calls = []

        def injected(inp, p=0.5, training=True, inplace=False):
            assert training and not inplace
            i = len(calls)
            if i == 0:
                site, kind = SITE_EMB, "emb"
            else:
                layer, k = divmod(i - 1, 4)
                site, kind = 8 * layer + (SITE_ATTN, SITE_OUT, SITE_FF1, SITE_FF2)[k], ("attn", "out", "ff1", "ff2")[k]
            calls.append((kind, tuple(inp.shape), p))
            if kind == "attn":
                b, h, n, _ = inp.shape
                keep = dropmask.keep_attention(b * h, n, p, SEED, site).reshape(inp.shape)
            else:
                keep = dropmask.keep_elementwise(inp.numel(), p, SEED, site).reshape(inp.shape)
            return inp * (torch.from_numpy(keep).float() / (1.0 - p))

        orig = F.dropout
        F.dropout = injected
        try:
            y = m(x)
        finally:
            F.dropout = orig
        assert len(calls) == 1 + 4 * depth and calls[1][0] == "attn" and len(calls[1][1]) == 4, calls
        out[tag] = y.numpy()
        out[tag + ".eval"] = m.eval()(x).numpy()
        print(tag, "train-vs-eval rel diff", float((y - m(x)).abs().max() / m(x).std()))
    np.savez(os.path.join(HERE, "f16_svit_train_drop.npz"), **out)
    print("wrote f16_svit_train_drop.npz", os.path.getsize(os.path.join(HERE, "f16_svit_train_drop.npz")), "bytes")


if __name__ == "__main__":
    main()
