"""CPU tier: the oracle (oracle/*.py, our from-scratch restatement) against the golden vectors that
tests/golden/make_golden.py produced by running the REFERENCE's own modules. Tolerances: integer
outputs exact; fp32 outputs 1e-5 relative to the tensor's std (different op order only)."""
import types

import numpy as np
import pytest
import torch

from oracle import ddim as oddim
from oracle import style as ostyle
from oracle import unet as ounet
from stedm_amd.utils import prng
from tests.golden.summary import check_summary

torch.set_grad_enabled(False)
TOL = 2e-5


def close(a, b, tol=TOL):
    a = torch.as_tensor(np.asarray(a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = float(b.std()) if b.numel() > 1 else float(b.abs().max())
    err = float((a - b).abs().max()) / (scale + 1e-12)
    assert err <= tol, f"max err/std {err:.3e} > {tol}"


def test_f1_timestep_embedding(golden):
    fx = golden("f1_timestep_embedding")
    emb = ounet.timestep_embedding(torch.from_numpy(fx["t"]), 128)
    close(emb, fx["emb"], 1e-6)


def test_f2_schedule(golden):
    fx = golden("f2_schedule")
    s = oddim.Schedule()
    assert np.array_equal(s.betas.numpy(), fx["betas_f32"])
    assert np.array_equal(s.alphas_cumprod.numpy(), fx["alphas_cumprod_f32"])
    for S, n in ((20, 20), (50, 50), (128, 143)):
        ts = oddim.make_ddim_timesteps(S)
        assert ts.shape[0] == n
        assert np.array_equal(ts.astype(np.int64), fx[f"ts_{S}"])  # integer: bit-exact
        for eta in (0.0, 1.0):
            sig, a, ap = oddim.make_ddim_sampling_parameters(s.alphas_cumprod, ts, eta)
            assert np.array_equal(np.asarray(sig, dtype=np.float64), fx[f"sig_{S}_{eta}"])
            assert np.array_equal(np.asarray(a, dtype=np.float64), fx[f"a_{S}_{eta}"])
            assert np.array_equal(np.asarray(ap, dtype=np.float64), fx[f"ap_{S}_{eta}"])
    # SURVEY.md §8(a) A11 spot values
    assert abs(float(s.alphas_cumprod[0]) - 9.984999895e-01) < 1e-9
    assert abs(float(s.alphas_cumprod[999]) - 9.691086598e-05) < 1e-12


def _res_params(cin, cout, edim, seed):
    S = {}
    S["in_layers.0.weight"] = (cin,); S["in_layers.0.bias"] = (cin,)
    S["in_layers.2.weight"] = (cout, cin, 3, 3); S["in_layers.2.bias"] = (cout,)
    S["emb_layers.1.weight"] = (cout, edim); S["emb_layers.1.bias"] = (cout,)
    S["out_layers.0.weight"] = (cout,); S["out_layers.0.bias"] = (cout,)
    S["out_layers.3.weight"] = (cout, cout, 3, 3); S["out_layers.3.bias"] = (cout,)
    if cin != cout:
        S["skip_connection.weight"] = (cout, cin, 1, 1); S["skip_connection.bias"] = (cout,)
    return prng.fill_state_dict(S, seed)


def test_f3_resblock_updown(golden):
    fx = golden("f3_resblock_updown")
    for tag, (cin, cout, hw, edim) in {"same": (64, 64, 8, 128), "skip": (64, 128, 8, 128), "wide": (256, 128, 16, 512)}.items():
        P = _res_params(cin, cout, edim, 3)
        x = prng.normal(3, f"rb.{tag}.x", (2, cin, hw, hw))
        e = prng.normal(3, f"rb.{tag}.emb", (2, edim))
        y = ounet.resblock(P, "", x, e)
        if tag == "wide":
            check_summary(y, fx, tag, TOL)
        else:
            close(y, fx[tag])
    P = prng.fill_state_dict({"op.weight": (64, 64, 3, 3), "op.bias": (64,)}, 5)
    y = ounet._run_ops(P, [("down", "", {})], prng.normal(5, "down.x", (2, 64, 16, 16)), None, None)
    close(y, fx["down"])
    P = prng.fill_state_dict({"conv.weight": (64, 64, 3, 3), "conv.bias": (64,)}, 5)
    y = ounet._run_ops(P, [("up", "", {})], prng.normal(5, "up.x", (2, 64, 8, 8)), None, None)
    close(y, fx["up"])


def test_f4_attention_block(golden):
    fx = golden("f4_attention_block")
    for tag, (c, heads, hw) in {"small": (128, 8, 8), "mid1024": (1024, 8, 8), "t256": (128, 4, 16)}.items():
        S = {"norm.weight": (c,), "norm.bias": (c,), "qkv.weight": (3 * c, c, 1), "qkv.bias": (3 * c,),
             "proj_out.weight": (c, c, 1), "proj_out.bias": (c,)}
        P = prng.fill_state_dict(S, 4)
        x = prng.normal(4, f"attn.{tag}.x", (1 if c == 1024 or hw == 16 else 2, c, hw, hw))
        y = ounet.attention_block(P, "", x, heads)
        if c == 1024:
            check_summary(y, fx, tag, TOL)
        else:
            close(y, fx[tag])


def test_f11_spatial_transformer(golden):
    fx = golden("f11_spatial_transformer")
    cfg = ounet.UNetConfig(model_channels=16, channel_mult=(1, 2, 8), num_heads=8, use_spatial_transformer=True, context_dim=128)
    plan = ounet.build_plan(cfg)  # middle width 128, 8 heads x 16
    pre = "middle_block.2."
    shapes = {k[len(pre):]: v for k, v in plan.shapes.items() if k.startswith(pre)}
    P = prng.fill_state_dict(shapes, 11)
    x = prng.normal(11, "st.x", (2, 128, 8, 8))
    close(ounet.spatial_transformer(P, "", x, None, 8, 1), fx["y"])


def _unet_case(fx, tag, B, hw, seed, cfg, taps=True):
    plan = ounet.build_plan(cfg)
    P = prng.fill_state_dict(plan.shapes, seed)
    assert sum(int(np.prod(s)) for s in plan.shapes.values()) == int(fx["n_params"])
    x = prng.normal(seed, f"unet.{tag}.x", (B, cfg.in_channels, hw, hw))
    ctx = prng.normal(seed, f"unet.{tag}.ctx", (B, cfg.model_channels * 4))
    tt = torch.from_numpy(fx["t"])
    tp = {} if taps else None
    y = ounet.unet_forward(P, cfg, x, tt, ctx, plan=plan, taps=tp)
    if "y" in fx.files:
        close(y, fx["y"], 5e-5)
    check_summary(y, fx, "y", 5e-5, tag)
    if taps:
        for k, v in tp.items():
            check_summary(v, fx, k, 5e-5, tag)


def test_f6_unet_tiny(golden):
    cfg = ounet.UNetConfig(image_size=16, in_channels=7, model_channels=32, out_channels=4, channel_mult=(1, 2, 4), num_heads=4)
    _unet_case(golden("f6_unet_tiny"), "tiny", 2, 16, 6, cfg)


def test_f6_unet_ns32(golden):
    _unet_case(golden("f6_unet_ns32"), "ns32", 2, 32, 0, ounet.UNetConfig())


def test_f7_svit(golden):
    fx = golden("f7_svit")
    for tag, (img, ns_, B) in {"i64_ns1": (64, 1, 2), "i64_ns4": (64, 4, 2)}.items():
        cfg = ostyle.SViTConfig(image_size=img, ns=ns_)
        P = prng.fill_state_dict(ostyle.svit_shapes(cfg), 7)
        for l in range(cfg.depth):
            P[f"transformer.layers.{l}.0.fn.temperature"] = torch.tensor(float(np.log(64 ** -0.5)) + 0.05 * l)
        x = prng.uniform(7, f"svit.{tag}.img", (B, ns_, img, img, 3))
        close(ostyle.svit_forward(P, cfg, x), fx[tag], 5e-5)


def test_philox_known_answers_and_mask_statistics():
    """The mask streams of the train-mode dropout (oracle/dropmask.py): Philox4x32-10 against Random123's known-answer vectors, keep rates of
    both stream kinds, and independence of sites / seeds."""
    from oracle import dropmask as dm
    for c, k, want in dm.KAT:
        got = dm.philox4x32_10(*[np.uint32(v) for v in c], k[0], k[1])
        assert tuple(int(v) for v in got) == want
    n = 1 << 20
    a = dm.keep_elementwise(n, 0.1, 1234, 3)
    assert abs(a.mean() - (1 - 6554 / 65536)) < 4 * np.sqrt(0.09 / n)
    assert dm.keep_elementwise(n, 0.0, 1234, 3).all()
    b = dm.keep_elementwise(n, 0.1, 1234, 4)
    c = dm.keep_elementwise(n, 0.1, 1235, 3)
    for other in (b, c):                                   # different site / seed: independent masks (joint drop rate ~ p^2)
        assert abs((~a & ~other).mean() - 0.01) < 2e-3
    assert (dm.keep_elementwise(1003, 0.1, 1234, 3) == a[:1003]).all()          # a prefix property of the linear index
    k = dm.keep_attention(4, 200, 0.1, 99, 1)
    assert k.shape == (4, 200, 200)
    assert abs(k.mean() - 0.9) < 4 * np.sqrt(0.09 / k.size)
    assert abs(k.mean(axis=(0, 1)).std()) < 0.02 and abs(k.mean(axis=(0, 2)).std()) < 0.03       # no key / query column stands out
    assert dm.keep_attention(4, 200, 0.0, 99, 1).all()
    assert (dm.keep_attention(2, 130, 0.1, 99, 1) == k[:2, :130, :130]).all()   # a tile's masks do not depend on T or on later sample-heads


def test_f16_svit_train_mode_dropout(golden):
    """The reference's sViT in train mode with the build's masks injected at every nn.Dropout (tests/golden/make_golden_train_drop.py): the
    oracle's train-mode path applies the same masks at the same sites with torch's arithmetic."""
    fx = golden("f16_svit_train_drop")
    seed = int(fx["seed"][0])
    for tag, (img, ns_, B, depth, heads) in {"i64_ns4_d2": (64, 4, 2, 2, 12), "i32_ns1_d3": (32, 1, 3, 3, 4)}.items():
        cfg = ostyle.SViTConfig(image_size=img, ns=ns_, depth=depth, heads=heads)
        P = prng.fill_state_dict(ostyle.svit_shapes(cfg), 7)
        for l in range(cfg.depth):
            P[f"transformer.layers.{l}.0.fn.temperature"] = torch.tensor(float(np.log(64 ** -0.5)) + 0.05 * l)
        x = prng.uniform(7, f"svit.train.{tag}.img", (B, ns_, img, img, 3))
        close(ostyle.svit_forward(P, cfg, x, train_drop=(0.1, 0.1, seed)), fx[tag], 5e-5)
        close(ostyle.svit_forward(P, cfg, x), fx[tag + ".eval"], 5e-5)


def test_f8_agg(golden):
    fx = golden("f8_agg")
    sty = prng.uniform(8, "agg.style", (2, 4, 16, 16, 3))

    def embedder_for(prefix):
        w = prng.fill_value(8, prefix + "proj.weight", (512, 48))
        b = prng.fill_value(8, prefix + "proj.bias", (512,))
        return lambda x: torch.nn.functional.linear(torch.nn.functional.adaptive_avg_pool2d(x, 4).flatten(1), w, b)

    close(ostyle.agg_mean(sty, embedder_for("embedder.")), fx["mean"])
    close(ostyle.agg_max(sty, embedder_for("embedder.")), fx["max"])
    P = prng.fill_state_dict({"linear_block.1.weight": (512, 2048), "linear_block.1.bias": (512,),
                              "linear_block.3.weight": (512, 512), "linear_block.3.bias": (512,)}, 8)
    close(ostyle.agg_linear(P, sty, embedder_for("embedder.")), fx["linear"])
    assert np.array_equal(ostyle.agg_none(sty).numpy(), fx["none"])


def test_f9_rescaler(golden):
    fx = golden("f9_rescaler")
    w = prng.fill_value(9, "channel_mapper.weight", (3, 2, 1, 1))
    seg = (prng.uniform(9, "resc.seg", (2, 2, 64, 64)) > 0).float()
    close(ostyle.spatial_rescaler(seg, w), fx["y"], 1e-6)
    # numerically == avg_pool2d(x, 4) then 1x1 conv
    alt = torch.nn.functional.conv2d(torch.nn.functional.avg_pool2d(seg, 4), w)
    close(alt, fx["y"], 1e-5)


def _toy(x, t, c):
    tf = t.float()[:, None, None, None] / 1000.0
    return torch.tanh(x * (0.5 + tf) + c["bias"]) * (0.8 + 0.3 * tf) + 0.1 * c["bias"]


def test_f10_ddim(golden):
    fx = golden("f10_ddim")
    sched = oddim.Schedule()
    xT = prng.normal(10, "ddim.xT", (2, 4, 8, 8))
    cond = {"bias": prng.normal(10, "ddim.c", (2, 4, 8, 8)) * 0.3}
    unc = {"bias": prng.normal(10, "ddim.u", (2, 4, 8, 8)) * 0.3}
    calls = [0]

    def am(x, t, c):
        calls[0] += 1
        return _toy(x, t, c)

    s = oddim.ddim_sample(am, sched, xT, cond, 20, 0.0, uncond=unc, scale=1.5)
    assert calls[0] == int(fx["cfg20_calls"]) == 40
    close(s, fx["cfg20"], 1e-5)
    calls[0] = 0
    s = oddim.ddim_sample(am, sched, xT, cond, 50, 0.0)
    assert calls[0] == int(fx["nocfg50_calls"]) == 50
    close(s, fx["nocfg50"], 1e-5)
    # single step, eta = 1, injected noise
    ds = oddim.DDIMSchedule(sched, 20, 1.0)
    t = torch.full((2,), 501, dtype=torch.long)
    e = oddim.cfg_combine(_toy(xT, t, cond), _toy(xT, t, unc), 1.5)
    xp, x0 = oddim.ddim_update(xT, e, *ds.scalars(10), noise=torch.from_numpy(fx["step_noise"]))
    close(xp, fx["step_xprev"], 1e-5)
    close(x0, fx["step_x0"], 1e-5)


@pytest.mark.parametrize("tag,B,hw,seed,cfg", [
    ("tiny", 2, 16, 6, dict(image_size=16, in_channels=7, model_channels=32, out_channels=4, channel_mult=(1, 2, 4), num_heads=4)),
    ("ns32", 2, 32, 0, {})])
def test_f14_gradients(golden, tag, B, hw, seed, cfg):
    """Reverse-mode gradients of the oracle (autograd over the functional restatement) against the reference's own
    UNetModel forward + L1 + backward (make_golden_grads.py): loss, dL/dx, dL/dcontext, every parameter's grad norm + samples."""
    from oracle import train as otrain
    from tests.golden.make_golden_grads import pick_index
    fx = golden(f"f14_grads_{tag}")
    cfg = ounet.UNetConfig(**cfg)
    plan = ounet.build_plan(cfg)
    P = prng.fill_state_dict(plan.shapes, seed)
    x = prng.normal(seed, f"unet.{tag}.x", (B, cfg.in_channels, hw, hw))
    ctx = prng.normal(seed, f"unet.{tag}.ctx", (B, cfg.model_channels * 4))
    target = prng.normal(seed, f"unet.{tag}.target", (B, cfg.out_channels, hw, hw))
    loss, grads, dx, dctx, _ = otrain.unet_loss_and_grads(P, cfg, x, torch.from_numpy(fx["t"]), ctx, target)
    assert abs(loss - float(fx["loss"])) < 1e-5 * float(fx["loss"])
    check_summary(dx, fx, "dx", 2e-4, tag)
    close(dctx, fx["dctx"], 2e-4)
    assert set(grads) == {k[2:-5] for k in fx.files if k.startswith("g.") and k.endswith(".norm")}
    for name, g in grads.items():
        n = float(fx[f"g.{name}.norm"])
        a = g.double().reshape(-1)
        assert abs(float(a.norm()) - n) <= 2e-4 * n + 1e-12, name
        pick = a[torch.from_numpy(pick_index(a.numel()))].numpy()
        assert np.abs(pick - fx[f"g.{name}.pick"]).max() <= 2e-4 * n / np.sqrt(a.numel()) * 30 + 1e-9, name


def test_f13_adamw_and_litema(golden):
    """oracle/train.py's AdamW and EMA restatements against traces of `torch.optim.AdamW` and of the reference's own `LitEma`
    (tests/golden/make_golden_opt.py imports ldm.modules.ema): parameters, both moments and the shadows after each of three steps,
    plus one update in the saturated branch of LitEma's decay schedule."""
    from oracle import train as otrain
    fx = golden("f13_ema_adamw")
    lr, b1, b2, eps, wd, dec = [float(v) for v in fx["hyper"]]
    names = sorted(k[3:] for k in fx.files if k.startswith("p0."))
    p = {n: torch.from_numpy(fx[f"p0.{n}"].copy()) for n in names}
    m = {n: torch.zeros_like(p[n]) for n in names}
    v = {n: torch.zeros_like(p[n]) for n in names}
    sh = {n: p[n].clone() for n in names}
    for step in (1, 2, 3):
        assert int(fx[f"num_updates{step}"]) == step
        d = otrain.ema_decay(step, dec)
        for n in names:
            g = prng.normal(13, f"f13.g{step}.{n}", tuple(p[n].shape)) * 0.3
            otrain.adamw_step(p[n], g, m[n], v[n], step, lr, b1, b2, eps, wd)
            otrain.ema_update(sh[n], p[n], d)
            for got, key in ((p[n], "p"), (m[n], "m"), (v[n], "v"), (sh[n], "ema")):
                ref = torch.from_numpy(fx[f"{key}{step}.{n}"])
                assert torch.allclose(got, ref, rtol=2e-6, atol=1e-7), (key, step, n, float((got - ref).abs().max()))
    nL = int(fx["num_updatesL"])
    assert nL == 200001 and otrain.ema_decay(nL, dec) == dec
    for n in names:
        otrain.ema_update(sh[n], torch.from_numpy(fx[f"pL.{n}"]), otrain.ema_decay(nL, dec))
        assert torch.allclose(sh[n], torch.from_numpy(fx[f"emaL.{n}"]), rtol=2e-6, atol=1e-7), n


@pytest.mark.parametrize("tag,cfgkw,B,side", [("tiny", dict(ch=32, num_res_blocks=1), 2, 64), ("f4", {}, 1, 128)])
def test_f15_vq_encoder_decoder(golden, tag, cfgkw, B, side):
    """oracle/vq.py Encoder / Decoder against the reference's own model.Encoder / model.Decoder (make_golden_vq.py); `f4` is the shipped
    vq-f4.yaml architecture (55.3 M parameters) at a 128^2 image."""
    from oracle import vq as ovq
    fx = golden(f"f15_vq_{tag}")
    cfg = ovq.VQConfig(**cfgkw)
    P = prng.fill_state_dict(ovq.shapes(cfg), 15)
    assert sum(v.numel() for k, v in P.items() if k.startswith("encoder.")) == int(fx["n_enc"])
    assert sum(v.numel() for k, v in P.items() if k.startswith("decoder.")) == int(fx["n_dec"])
    x = prng.uniform(15, f"vq.{tag}.x", (B, 3, side, side))
    z = prng.normal(15, f"vq.{tag}.z", (B, 3, side // 4, side // 4))
    close(ovq.encoder(P, cfg, x), fx["enc_out"], 5e-5)
    y = ovq.decoder(P, cfg, z)
    check_summary(y, fx, "dec_out", 5e-5, tag)
    if "dec_out" in fx.files:
        close(y, fx["dec_out"], 5e-5)


def test_vq_quantiser_restatement_properties():
    """The quantiser (taming VectorQuantizer2; parity unpinned) restated with pinned-down fp32 arithmetic: the chosen entry is a nearest
    one under exact arithmetic up to rounding, ties go to the first index, and the straight-through value is z + (e - z)."""
    from oracle import vq as ovq
    cb = prng.normal(16, "vq.cb", (512, 3)) * 0.5
    z = prng.normal(16, "vq.z", (2, 3, 8, 8))
    idx, zq = ovq.quantize(cb, z)
    zf = z.permute(0, 2, 3, 1).reshape(-1, 3).double()
    d = ((zf[:, None, :] - cb.double()[None]) ** 2).sum(-1)
    best = d.min(dim=1).values
    assert float((d[torch.arange(d.shape[0]), idx.reshape(-1)] - best).max()) < 1e-5
    e = cb[idx.reshape(-1)]
    assert torch.equal(zq.permute(0, 2, 3, 1).reshape(-1, 3), zf.float() + (e - zf.float()))
    cb2 = torch.cat([cb, cb[:4]])                        # duplicated entries: the first occurrence must win
    idx2, _ = ovq.quantize(cb2, z)
    assert torch.equal(idx, idx2)
