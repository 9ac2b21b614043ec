"""GPU tier (-m gpu): Swin-Transformer-V2 style embedder — the HIP-backed `stedm_amd.swin.SwinTransformerV2` against the
CPU oracle `oracle/swin.py` (restated torchvision algorithm; PARITY UNPINNED: torchvision is absent, see the oracle's header).
Edge cases follow torchvision's own: feature maps that are not multiples of the window (F.pad rows take part as keys), a side the window
covers (no shift along it), odd sides in PatchMergingV2."""
import math

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from stedm_amd.utils import prng

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def rel(a, b):
    a = a.double().cpu(); b = torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max()) / (float(b.std()) + 1e-12)


def fill_swin_(m, seed=11):
    """PRNG recipe for the Swin containers: LayerNorm affine 1 + 0.1 N / 0.1 N, logit_scale log(10) + 0.3 N, biases 0.05 N (the key third of
    qkv.bias too: the forward must zero it), weights N(0, 1/sqrt(fan_in))."""
    for mod_name, mod in m.named_modules():
        for pn, p in mod.named_parameters(recurse=False):
            name = f"{mod_name}.{pn}" if mod_name else pn
            if isinstance(mod, nn.LayerNorm):
                p.copy_(prng.normal(seed, name, p.shape, std=0.1, mean=1.0 if pn == "weight" else 0.0))
            elif pn == "logit_scale":
                p.copy_(prng.normal(seed, name, p.shape, std=0.3, mean=math.log(10.0)))
            elif p.dim() >= 2:
                p.copy_(prng.normal(seed, name, p.shape, std=1.0 / math.sqrt(int(np.prod(p.shape[1:])))))
            else:
                p.copy_(prng.normal(seed, name, p.shape, std=0.05))
    return m


def make_swin(dev, precision="parity", classes=512, **kw):
    from stedm_amd.swin import swin_v2_t
    m = swin_v2_t(num_classes=classes, precision=precision, **kw).eval()
    fill_swin_(m)
    params = {k: v.clone() for k, v in m.state_dict().items()}
    return m.to(dev), params


@pytest.mark.parametrize("precision", ["bf16", "f16"])
@pytest.mark.parametrize("M,K,N,res", [(70000, 128, 96, True), (70000, 384, 96, True), (33333, 64, 96, False), (40000, 128, 128, True), (50001, 192, 64, False)])
def test_gemm_with_layernorm_epilogue(dev, precision, M, K, N, res):
    """x = (res +) LayerNorm(a @ w^T + bias) in the 1x1 GEMM's epilogue (stedm_conv_args.ln_*; Swin-V2's res-post-norm at 96 channels,
    torchvision swin_transformer.py SwinTransformerBlockV2.forward) against the GEMM's fp32 output + torch's layer_norm; in place on the
    residual stream as the embedder calls it; ragged last M-tile."""
    from stedm_amd import ops
    from stedm_amd._lib import F16
    prec = ops.Precision.parse(precision)
    ft = torch.float16 if prec.mm_dtype == F16 else torch.bfloat16
    x16 = (torch.randn(M, K, device=dev) * 0.7).to(ft).view(torch.int16)
    w = torch.randn(N, K, 1, 1, device=dev) / math.sqrt(K)
    bs = torch.randn(N, device=dev) * 0.3
    g = torch.randn(N, device=dev) * 0.2 + 1.0; b = torch.randn(N, device=dev) * 0.1
    r0 = torch.randn(M, N, device=dev) if res else None
    whi, wlo = ops.pack_conv_weight(w, prec); wf = ops.pack_conv_weight_frag(w, prec)
    v4 = lambda t: t.view(1, 1, M, -1)
    y = torch.empty(M, N, device=dev)
    ops.conv_igemm(None, whi, wlo, v4(y), prec=prec, ks=1, src16=(v4(x16), None), w_frag=wf, bias=bs)
    ref = torch.nn.functional.layer_norm(y.double(), (N,), g.double(), b.double(), 1e-5)
    if res: ref = ref + r0.double()
    xc = r0.clone() if res else torch.full((M, N), float("nan"), device=dev)
    h16 = torch.full((M, N), 0x7FFF, dtype=torch.int16, device=dev)
    kw = dict(prec=prec, ks=1, src16=(v4(x16), None), w_frag=wf, bias=bs, out16=(v4(h16), None), ln_after=(g, b, 1e-5, xc if res else None))
    assert ops.conv_igemm(None, whi, wlo, v4(xc), query_rs=True, **kw)
    ops.conv_igemm(None, whi, wlo, v4(xc), **kw)
    err = float((xc.double() - ref).abs().max()) / float(ref.std())
    assert err < 1e-5, err
    assert torch.equal(h16.view(ft), xc.to(ft))


def test_swin_rpb_and_buffers_match_oracle(dev):
    from oracle import swin as osw
    from stedm_amd import ops
    from stedm_amd.swin import ShiftedWindowAttentionV2
    at = ShiftedWindowAttentionV2(96, [8, 8], [4, 4], 3)
    assert torch.equal(at.relative_position_index.view(64, 64), osw.relative_position_index())
    assert torch.allclose(at.relative_coords_table.view(-1, 2), osw.relative_coords_table(), atol=0, rtol=0)
    fill_swin_(at, 3)
    p = {k: v.clone() for k, v in at.state_dict().items()}
    want = osw.position_bias(p, "", 3)                                   # [heads, query, key]
    at = at.to(dev)
    l0, l2 = at.cpb_mlp[0], at.cpb_mlp[2]
    table = at.relative_coords_table.reshape(-1, 2).contiguous()
    h1 = ops.linear(table, ops.transpose(l0.weight), l0.bias, torch.empty((225, 512), device=dev), act_out=2)
    cpb = ops.linear(h1, ops.transpose(l2.weight), None, torch.empty((225, 3), device=dev))
    got = ops.swin_rpb(cpb, at.relative_position_index, 3)
    assert float((got.cpu() - want).abs().max()) < 2e-5


@pytest.mark.parametrize("H,W,shift", [(16, 16, 0), (16, 16, 4), (12, 20, 4), (8, 24, 4), (4, 4, 4), (24, 8, 4), (9, 17, 4)])
def test_swin_window_attention_kernel_vs_oracle(dev, H, W, shift):
    """cosine window attention incl. roll, partition, F.pad rows, mask of a partly shifted map; `proj` is the identity here."""
    from oracle import swin as osw
    from stedm_amd import ops
    from stedm_amd.ops import Precision
    from stedm_amd.swin import ShiftedWindowAttentionV2
    heads, C, N = 3, 96, 2
    at = fill_swin_(ShiftedWindowAttentionV2(C, [8, 8], [shift, shift], heads), 5)
    p = {k: v.clone() for k, v in at.state_dict().items()}
    p["proj.weight"], p["proj.bias"] = torch.eye(C), torch.zeros(C)
    x = prng.normal(5, f"swin.attn.{H}.{W}", (N, H, W, C))
    want = osw.window_attention(x, p, "", heads, shift)
    bz = p["qkv.bias"].clone(); bz[C:2 * C] = 0
    qkv = F.linear(x, p["qkv.weight"], bz).reshape(N * H * W, 3 * C).contiguous().to(dev)
    rpb = osw.position_bias(p, "", heads).contiguous().to(dev)
    scale = torch.clamp(p["logit_scale"].reshape(-1), max=math.log(100.0)).exp().to(dev)
    prec = Precision.parse("parity")
    hi = torch.zeros((N * H * W, C), dtype=torch.int16, device=dev)
    lo = torch.zeros_like(hi)
    ops.swin_window_attn(qkv, bz.to(dev), scale, rpb, hi, lo, N, H, W, heads, shift, prec)
    got = (hi.view(torch.float16).float() + lo.view(torch.float16).float()).view(N, H, W, C)
    err = rel(got, want)
    print(f"[swin window attention {H}x{W} shift {shift}] {err:.2e}")
    assert err < 2e-5


def test_swin_gather_kernels_vs_torch(dev):
    from stedm_amd import ops
    from stedm_amd.ops import Precision
    prec = Precision.parse("parity")
    f16 = lambda hi, lo: hi.view(torch.float16).float() + lo.view(torch.float16).float()
    # patch rows from a permuted NHWC view (what Agg_* hands over)
    img = prng.uniform(3, "swin.img", (2, 24, 40, 3)).to(dev)
    x = img.permute(0, 3, 1, 2)
    hi = torch.empty((2 * 6 * 10, 64), dtype=torch.int16, device=dev); lo = torch.empty_like(hi)
    ops.swin_patch16(x, hi, lo, prec)
    want = F.unfold(x.contiguous(), kernel_size=4, stride=4).transpose(1, 2).reshape(-1, 48)      # (c, ky, kx) columns, row-major patches
    got = f16(hi, lo)
    assert float((got[:, :48] - want).abs().max()) < 1e-6 and float(got[:, 48:].abs().max()) == 0.0
    # 2x2 merge with odd sides
    t = prng.normal(3, "swin.merge", (2, 5, 7, 32)).to(dev)
    hi = torch.empty((2 * 3 * 4, 128), dtype=torch.int16, device=dev); lo = torch.empty_like(hi)
    ops.swin_merge16(t, hi, lo, prec)
    tp = F.pad(t, (0, 0, 0, 1, 0, 1))
    want = torch.cat([tp[:, 0::2, 0::2], tp[:, 1::2, 0::2], tp[:, 0::2, 1::2], tp[:, 1::2, 1::2]], -1).reshape(-1, 128)
    assert float((f16(hi, lo) - want).abs().max()) < 1e-6
    # residual LayerNorm (in place) + planes, token mean
    y = prng.normal(3, "swin.ln.y", (50, 96)).to(dev); r = prng.normal(3, "swin.ln.r", (50, 96)).to(dev)
    g = prng.normal(3, "swin.ln.g", (96,), 0.1, 1.0).to(dev); b = prng.normal(3, "swin.ln.b", (96,), 0.1).to(dev)
    want = r + F.layer_norm(y, (96,), g, b, 1e-5)
    out = r.clone(); hi = torch.empty((50, 96), dtype=torch.int16, device=dev); lo = torch.empty_like(hi)
    ops.swin_ln(y, g, b, 1e-5, out, out, hi, lo, prec)
    assert float((out - want).abs().max()) < 1e-5 and float((f16(hi, lo) - out).abs().max()) < 1e-6
    tm = ops.swin_token_mean(want.view(2, 25, 96).contiguous(), torch.empty((2, 96), device=dev))
    assert float((tm - want.view(2, 25, 96).mean(1)).abs().max()) < 1e-6


@pytest.mark.parametrize("tag,N,H,W", [("256", 2, 256, 256), ("ragged", 2, 144, 208), ("512", 1, 512, 512)])
def test_swin_v2_t_vs_oracle(dev, tag, N, H, W):
    """Whole embedder with the reference's head (Linear(768, 512), s_zss_dm.py:20), parity mode, at 1e-3. 512 x 512 is the reference's patch
    size (all four stages shifted); 256: the last stage is one window (no shift); ragged: padded windows and odd merges."""
    from oracle import swin as osw
    m, params = make_swin(dev)
    x = prng.uniform(11, f"swin.img.{tag}", (N, H, W, 3))
    want = osw.swin_v2_forward(params, x.permute(0, 3, 1, 2).contiguous())
    got = m(x.to(dev).permute(0, 3, 1, 2))            # the strided '(b n) c h w' view of NHWC images
    err = rel(got, want)
    print(f"[swin_v2_t {tag}] parity-mode max|diff|/std vs CPU oracle: {err:.3e}")
    assert err < 1e-3


@pytest.mark.parametrize("precision,tol", [("f16", 2e-2), ("bf16", 1e-1)])
def test_swin_fast_modes_reported(dev, precision, tol):
    from oracle import swin as osw
    m, params = make_swin(dev, precision)
    x = prng.uniform(11, "swin.img.256", (2, 256, 256, 3))
    want = osw.swin_v2_forward(params, x.permute(0, 3, 1, 2).contiguous())
    err = rel(m(x.to(dev).permute(0, 3, 1, 2)), want)
    print(f"[swin_v2_t 256 {precision}] max|diff|/std vs CPU oracle: {err:.3e}")
    assert err < tol


def test_swin_chunking(dev):
    """images are independent: any chunking gives the same rows (to rounding: the small-batch Linear / split-K forms sum in another order)."""
    m, _ = make_swin(dev, chunk_images=2)
    x = prng.uniform(11, "swin.img.chunk", (5, 3, 64, 64)).to(dev)
    a = m(x).clone()
    m.chunk_images = 8
    assert rel(a, m(x).cpu()) < 1e-5


def test_agg_blocks_with_hip_embedder(dev):
    """agg_blocks.py:24-75 end to end on the HIP embedder: '(b n) c h w' -> swin -> mean / max / MLP."""
    from types import SimpleNamespace
    from oracle import swin as osw
    from stedm_amd.style import Agg_Linear, Agg_Max, Agg_Mean
    emb, params = make_swin(dev)
    imgs = prng.uniform(11, "swin.agg.img", (2, 3, 128, 128, 3))
    f = osw.swin_v2_forward(params, imgs.reshape(6, 128, 128, 3).permute(0, 3, 1, 2).contiguous()).view(2, 3, 512)
    cfg = SimpleNamespace(name="mp", num_patches=3)
    assert rel(Agg_Mean(cfg, emb)(imgs.to(dev)), f.mean(1)) < 1e-3
    assert rel(Agg_Max(cfg, emb)(imgs.to(dev)), f.max(1)[0]) < 1e-3
    lin = Agg_Linear(cfg, emb)
    prng.fill_module_(lin._linear_block, seed=13)
    lb = lin._linear_block
    want = F.relu(F.linear(F.relu(F.linear(F.relu(f.reshape(2, -1)), lb[1].weight, lb[1].bias)), lb[3].weight, lb[3].bias))
    lin = lin.to(dev)
    assert rel(lin(imgs.to(dev)), want) < 1e-3


# ------------------------------------------------------------------------------------------------ train mode: stochastic depth
def test_swin_train_mode_stochastic_depth_vs_oracle_with_the_same_gates(dev):
    """torchvision's swin_v2_t in train mode (the reference runs the embedder inside the training step: s_zss_dm.py:45-60) applies
    StochasticDepth(p_i, "row") to both residual branches of block i, p_i = 0.2 i / 11: per-image gates bernoulli(1 - p) / (1 - p). torch's
    draw cannot be matched, so the gates are injected on both sides."""
    from oracle import swin as osw
    m, params = make_swin(dev)
    assert len(m.sd_probs) == 12 and m.sd_probs[0] == 0.0 and abs(m.sd_probs[-1] - 0.2) < 1e-12 and abs(m.sd_probs[5] - 0.2 * 5 / 11) < 1e-12
    N = 5
    g = torch.Generator().manual_seed(3)
    surv = 1.0 - torch.tensor(m.sd_probs).view(-1, 1, 1)
    gates = (torch.rand(12, 2, N, generator=g) < 0.6).float() / surv        # (a high drop rate so that every block sees both outcomes)
    gates[0] = 1.0                                                           # p_0 = 0: torchvision returns the input unchanged
    x = prng.uniform(11, "swin.img.train", (N, 64, 96, 3))
    want = osw.swin_v2_forward(params, x.permute(0, 3, 1, 2).contiguous(), sd_gates=gates)
    m.train()
    m.sd_gates = gates
    m.chunk_images = 2                                                       # gates follow the images through the chunks
    got = m(x.to(dev).permute(0, 3, 1, 2))
    err = rel(got, want)
    ev = osw.swin_v2_forward(params, x.permute(0, 3, 1, 2).contiguous())
    print(f"[swin_v2_t train] parity-mode max|diff|/std vs CPU oracle with the same gates: {err:.3e} (eval differs by {rel(got, ev):.2f})")
    assert err < 1e-3 and rel(got, ev) > 0.05
    # gates of one = eval arithmetic bit for bit
    m.sd_gates = torch.ones(12, 2, N)
    a = m(x.to(dev).permute(0, 3, 1, 2)).clone()
    assert torch.equal(a, m.eval()(x.to(dev).permute(0, 3, 1, 2)))


def test_swin_train_mode_draws_its_gates(dev):
    m, _ = make_swin(dev)
    m.train()
    x = prng.uniform(11, "swin.img.train2", (64, 3, 32, 32)).to(dev)
    torch.manual_seed(0)
    a = m(x).clone()
    g = m._gates.cpu()
    assert tuple(g.shape) == (12, 2, 64)
    for i, p in enumerate(m.sd_probs):                                       # values 0 or 1 / (1 - p_i); about p_i of them dropped
        for v in g[i].unique().tolist():
            assert v == 0.0 or abs(v - 1.0 / (1.0 - p)) < 1e-5, (i, v)
    assert abs(float((g[6:] == 0).float().mean()) - float(np.mean(m.sd_probs[6:]))) < 0.06
    assert bool(torch.isfinite(a).all()) and not torch.equal(a, m.eval()(x))
