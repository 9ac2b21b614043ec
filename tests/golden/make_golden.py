#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference's own modules
from /root/reference (read-only) in the build container, feeding them the PRNG recipe of
stedm_amd/utils/prng.py, and storing outputs (or compact summaries of large outputs).

Nothing of the reference travels: only small numeric arrays are written. Re-run with
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
The fixtures pin oracle/ (tests/test_oracle_golden.py) and, through it, the HIP path.
"""
from __future__ import annotations

import os
import sys
import types

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

torch.manual_seed(0)
torch.set_grad_enabled(False)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} KB")


def put_summary(d, key, t):
    for k, v in summarize(t).items():
        d[f"{key}.{k}"] = v


def main():
    from ldm.modules.diffusionmodules import util as rutil
    from ldm.modules.diffusionmodules import openaimodel as rom
    from ldm.modules import attention as ratt
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.modules.encoders.modules import SpatialRescaler
    from networks.vit_set import sViT
    from networks import agg_blocks as ragg

    # ---------------------------------------------------------------- F1 timestep embedding
    t = torch.tensor([0, 1, 500, 999, 951, 21], dtype=torch.long)
    save("f1_timestep_embedding", t=t.numpy(), emb=rutil.timestep_embedding(t, 128).numpy())

    # ---------------------------------------------------------------- F2 schedules (int parts exact)
    betas = rutil.make_beta_schedule("linear", 1000, linear_start=0.0015, linear_end=0.0205)
    ac = np.cumprod(1.0 - betas, axis=0)
    out = {"betas_f32": torch.tensor(betas, dtype=torch.float32).numpy(),
           "alphas_cumprod_f32": torch.tensor(ac, dtype=torch.float32).numpy()}
    acf = torch.tensor(ac, dtype=torch.float32)
    for S in (20, 50, 128):
        ts = rutil.make_ddim_timesteps("uniform", S, 1000, verbose=False)
        out[f"ts_{S}"] = ts.astype(np.int64)
        for eta in (0.0, 1.0):
            sig, a, ap = rutil.make_ddim_sampling_parameters(acf, ts, eta, verbose=False)
            out[f"sig_{S}_{eta}"] = np.asarray(sig, dtype=np.float64)
            out[f"a_{S}_{eta}"] = np.asarray(a, dtype=np.float64)
            out[f"ap_{S}_{eta}"] = np.asarray(ap, dtype=np.float64)
    save("f2_schedule", **out)

    # ---------------------------------------------------------------- F3 ResBlocks / F5 up+down
    out = {}
    for tag, (cin, cout, hw, edim) in {"same": (64, 64, 8, 128), "skip": (64, 128, 8, 128), "wide": (256, 128, 16, 512)}.items():
        m = rom.ResBlock(cin, edim, 0, out_channels=cout).eval()
        prng.fill_module_(m, seed=3)
        x = prng.normal(3, f"rb.{tag}.x", (2, cin, hw, hw))
        e = prng.normal(3, f"rb.{tag}.emb", (2, edim))
        y = m(x, e)
        if tag == "wide":
            put_summary(out, tag, y)
        else:
            out[tag] = y.numpy()
    m = rom.Downsample(64, True).eval(); prng.fill_module_(m, seed=5)
    out["down"] = m(prng.normal(5, "down.x", (2, 64, 16, 16))).numpy()
    m = rom.Upsample(64, True).eval(); prng.fill_module_(m, seed=5)
    out["up"] = m(prng.normal(5, "up.x", (2, 64, 8, 8))).numpy()
    save("f3_resblock_updown", **out)

    # ---------------------------------------------------------------- F4 AttentionBlock
    out = {}
    for tag, (c, heads, hw) in {"small": (128, 8, 8), "mid1024": (1024, 8, 8), "t256": (128, 4, 16)}.items():
        m = rom.AttentionBlock(c, num_heads=heads, num_head_channels=-1).eval()
        prng.fill_module_(m, seed=4)
        x = prng.normal(4, f"attn.{tag}.x", (1 if c == 1024 or hw == 16 else 2, c, hw, hw))
        y = m(x)
        if c == 1024:
            put_summary(out, tag, y)
        else:
            out[tag] = y.numpy()
    save("f4_attention_block", **out)

    # ---------------------------------------------------------------- F11 SpatialTransformer
    m = ratt.SpatialTransformer(128, 8, 16, depth=1, context_dim=128).eval()
    prng.fill_module_(m, seed=11)
    x = prng.normal(11, "st.x", (2, 128, 8, 8))
    save("f11_spatial_transformer", y=m(x).numpy())

    # ---------------------------------------------------------------- F6 U-Nets
    def run_unet(tag, B, hw, seed, hooks=True, **kw):
        m = rom.UNetModel(**kw).eval()
        prng.fill_module_(m, seed=seed)
        x = prng.normal(seed, f"unet.{tag}.x", (B, kw["in_channels"], hw, hw))
        ctx = prng.normal(seed, f"unet.{tag}.ctx", (B, kw["model_channels"] * 4))
        tt = torch.tensor(([951, 21, 500, 1] * B)[:B], dtype=torch.long)
        taps = {}
        hs = []
        if hooks:
            for i, blk in enumerate(m.input_blocks):
                hs.append(blk.register_forward_hook(lambda mod, a, o, i=i: taps.__setitem__(f"in{i}", o)))
            hs.append(m.middle_block.register_forward_hook(lambda mod, a, o: taps.__setitem__("mid", o)))
            for i, blk in enumerate(m.output_blocks):
                hs.append(blk.register_forward_hook(lambda mod, a, o, i=i: taps.__setitem__(f"out{i}", o)))
        y = m(x, tt, context=ctx)
        d = {"t": tt.numpy(), "n_params": np.int64(sum(p.numel() for p in m.parameters()))}
        if y.numel() <= 16384:
            d["y"] = y.numpy()
        put_summary(d, "y", y)
        for k, v in taps.items():
            put_summary(d, k, v)
        save(f"f6_unet_{tag}", **d)

    tiny = dict(image_size=16, in_channels=7, model_channels=32, out_channels=4, num_res_blocks=2,
                attention_resolutions=[32, 16, 8], channel_mult=[1, 2, 4], num_heads=4)
    run_unet("tiny", 2, 16, 6, **tiny)
    # NOTE: UNetModel(use_spatial_transformer=True, context_dim=...) imports omegaconf (openaimodel.py:499), which this
    # image lacks (ordinary ModuleNotFoundError) -> the ST-in-U-Net wiring is pinned by F11 (module) + text only.
    ns = dict(in_channels=7, model_channels=128, out_channels=4, num_res_blocks=2,
              attention_resolutions=[32, 16, 8], channel_mult=[1, 4, 8], num_heads=8)
    run_unet("ns32", 2, 32, 0, image_size=32, **ns)
    ref128 = dict(image_size=128, in_channels=6, model_channels=128, out_channels=3, num_res_blocks=2,
                  attention_resolutions=[32, 16, 8], channel_mult=[1, 4, 8], num_heads=8)
    run_unet("ref128", 1, 128, 0, hooks=False, **ref128)

    # ---------------------------------------------------------------- F7 sViT
    out = {}
    for tag, (img, ns_, B) in {"i64_ns1": (64, 1, 2), "i64_ns4": (64, 4, 2), "i512_ns4": (512, 4, 1)}.items():
        m = sViT(image_size=img, patch_size=8, num_classes=512, dim=256, depth=6, heads=12, mlp_dim=256, pool="mean",
                 channels=3, dropout=0.1, emb_dropout=0.1, ns=ns_, t_dim=256).eval()
        prng.fill_module_(m, seed=7)
        for l, (attn, _ff) in enumerate(m.transformer.layers):
            attn.fn.temperature.fill_(float(np.log(64 ** -0.5)) + 0.05 * l)
        x = prng.uniform(7, f"svit.{tag}.img", (B, ns_, img, img, 3))
        out[tag] = m(x).numpy()
    save("f7_svit", **out)

    # ---------------------------------------------------------------- F8 aggregation blocks (linear stand-in embedder)
    class StandIn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.proj = torch.nn.Linear(3 * 4 * 4, 512)

        def forward(self, x):  # [(b n), 3, H, W] -> [(b n), 512]
            return self.proj(torch.nn.functional.adaptive_avg_pool2d(x, 4).flatten(1))

    out = {}
    samp = types.SimpleNamespace(name="mp", num_patches=4)
    sty = prng.uniform(8, "agg.style", (2, 4, 16, 16, 3))
    for name, cls in (("mean", ragg.Agg_Mean), ("max", ragg.Agg_Max), ("linear", ragg.Agg_Linear)):
        emb = StandIn().eval()
        m = cls(samp, emb).eval()
        # state-dict aliasing (agg_blocks.py:21-22): fill via the registered names only
        sd = {k: prng.fill_value(8, k, v.shape) for k, v in m.state_dict().items() if not k.startswith("_")}
        m.load_state_dict(sd, strict=False)
        out[name] = m(sty).numpy()
    out["none"] = ragg.Agg_None(samp, None)(sty).numpy()
    save("f8_agg", **out)

    # ---------------------------------------------------------------- F9 SpatialRescaler
    m = SpatialRescaler(n_stages=2, in_channels=2, out_channels=3).eval()
    prng.fill_module_(m, seed=9)
    seg = (prng.uniform(9, "resc.seg", (2, 2, 64, 64)) > 0).float()
    save("f9_rescaler", y=m(seg).numpy())

    # ---------------------------------------------------------------- F10 DDIM sampler with a closed-form eps model
    class CPUSampler(DDIMSampler):
        def register_buffer(self, name, attr):  # harness override: the original pins "cuda" (ddim.py:18-22)
            setattr(self, name, attr)

    class Toy:
        """Duck-typed model surface the sampler reads (ddim.py:15,27-33,119)."""
        def __init__(self):
            acf_ = torch.tensor(ac, dtype=torch.float32)
            self.num_timesteps = 1000
            self.betas = torch.tensor(betas, dtype=torch.float32)
            self.alphas_cumprod = acf_
            self.alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32)
            self.device = torch.device("cpu")
            self.calls = 0

        def apply_model(self, x, t, c):
            self.calls += 1
            tf = t.float()[:, None, None, None] / 1000.0
            return torch.tanh(x * (0.5 + tf) + c["bias"]) * (0.8 + 0.3 * tf) + 0.1 * c["bias"]

    xT = prng.normal(10, "ddim.xT", (2, 4, 8, 8))
    cond = {"bias": prng.normal(10, "ddim.c", (2, 4, 8, 8)) * 0.3}
    unc = {"bias": prng.normal(10, "ddim.u", (2, 4, 8, 8)) * 0.3}
    out = {}
    toy = Toy()
    smp = CPUSampler(toy)
    s, _ = smp.sample(20, 2, (4, 8, 8), cond, verbose=False, eta=0.0, x_T=xT,
                      unconditional_guidance_scale=1.5, unconditional_conditioning=unc)
    out["cfg20"] = s.numpy(); out["cfg20_calls"] = np.int64(toy.calls)
    toy = Toy(); smp = CPUSampler(toy)
    s, _ = smp.sample(50, 2, (4, 8, 8), cond, verbose=False, eta=0.0, x_T=xT)
    out["nocfg50"] = s.numpy(); out["nocfg50_calls"] = np.int64(toy.calls)
    # one p_sample_ddim step, eta=1, with the global torch RNG pinned so the noise can be replayed
    toy = Toy(); smp = CPUSampler(toy)
    smp.make_schedule(20, ddim_eta=1.0, verbose=False)
    torch.manual_seed(1234)
    xp, x0 = smp.p_sample_ddim(xT, cond, torch.full((2,), 501, dtype=torch.long), index=10,
                               unconditional_guidance_scale=1.5, unconditional_conditioning=unc)
    torch.manual_seed(1234)
    out["step_noise"] = torch.randn(xT.shape).numpy()
    out["step_xprev"] = xp.numpy(); out["step_x0"] = x0.numpy()
    save("f10_ddim", **out)


if __name__ == "__main__":
    main()
