"""GPU tier (-m gpu): every HIP kernel, called through the C ABI, against the CPU oracle / plain
fp32 torch ops on the same seeded inputs. Tolerances (relative to the reference tensor's std):
  parity mode (fp16 x3 split products, fp32 accumulate)  : 2e-4 per op
  fast modes  (single fp16 / bf16 product)                : 5e-3 / 3e-2 per op (reported, loose)
  pure-fp32 kernels (GN, embeddings, attention, DDIM, I/O convs): 2e-5 .. 1e-4
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from stedm_amd.utils import prng

pytestmark = pytest.mark.gpu

torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from stedm_amd import _lib
    _lib.lib()  # must load: no fallback
    return torch.device("cuda:0")


def rel_err(got, ref):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float((got - ref).abs().max()) / (float(ref.std()) + 1e-12)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


PRECS = [("parity", 2e-4), ("f16", 5e-3), ("bf16", 3e-2)]


# ------------------------------------------------------------------------------------------------ GroupNorm
@pytest.mark.parametrize("B,H,W,c1,c2,bmod", [(2, 8, 8, 64, 0, 0), (3, 16, 16, 128, 0, 0), (2, 8, 8, 1024, 512, 0),
                                              (2, 4, 4, 32, 64, 0), (4, 8, 8, 128, 128, 2), (2, 32, 32, 128, 0, 0)])
def test_gn_scale_shift(dev, B, H, W, c1, c2, bmod):
    from stedm_amd import ops
    C = c1 + c2
    x1 = prng.normal(1, "gn.x1", (B, c1, H, W)) * 1.7 + 0.3
    x2 = prng.normal(1, "gn.x2", (bmod or B, c2, H, W)) * 0.6 - 0.2 if c2 else None
    g = prng.normal(1, "gn.g", (C,), 0.1, 1.0)
    b = prng.normal(1, "gn.b", (C,), 0.1)
    x2f = None if x2 is None else (x2 if not bmod else x2.repeat(B // bmod, 1, 1, 1))
    xin = x1 if x2 is None else torch.cat([x1, x2f], 1)
    ref = F.group_norm(xin, 32, g, b, 1e-5)
    sc = torch.empty((B, C), device=dev)
    sh = torch.empty((B, C), device=dev)
    ops.gn_scale_shift(nhwc(x1).to(dev), None if x2 is None else nhwc(x2).to(dev), g.to(dev), b.to(dev), 1e-5, sc, sh, 32, bmod)
    got = xin * sc.cpu()[:, :, None, None] + sh.cpu()[:, :, None, None]
    assert rel_err(got, ref) < 2e-5


# ------------------------------------------------------------------------------------------------ conv_igemm
def _conv_case(dev, prec_name, tol, B, Hin, Win, c1, c2, cout, mode, ks, gn, act, use_emb, use_res, bmod=0, seed=2):
    from stedm_amd import ops
    from stedm_amd._lib import CONV_DOWN, CONV_S1, CONV_UP
    prec = ops.Precision.parse(prec_name)
    cin = c1 + c2
    x1 = prng.normal(seed, "cv.x1", (B, c1, Hin, Win))
    x2 = prng.normal(seed, "cv.x2", (bmod or B, c2, Hin, Win)) if c2 else None
    w = prng.normal(seed, "cv.w", (cout, cin, ks, ks), 1.0 / math.sqrt(cin * ks * ks))
    bias = prng.normal(seed, "cv.b", (cout,), 0.05)
    x2f = None if x2 is None else (x2 if not bmod else x2.repeat(B // bmod, 1, 1, 1))
    xin = x1 if x2 is None else torch.cat([x1, x2f], 1)
    sc = sh = None
    a = xin
    if gn:
        sc = prng.normal(seed, "cv.sc", (B, cin), 0.2, 1.0)
        sh = prng.normal(seed, "cv.sh", (B, cin), 0.2)
        a = a * sc[:, :, None, None] + sh[:, :, None, None]
    if act:
        a = F.silu(a)
    if mode == "s1":
        ref = F.conv2d(a, w, bias, padding=ks // 2)
        m = CONV_S1
    elif mode == "down":
        ref = F.conv2d(a, w, bias, stride=2, padding=1)
        m = CONV_DOWN
    else:
        ref = F.conv2d(F.interpolate(a, scale_factor=2, mode="nearest"), w, bias, padding=1)
        m = CONV_UP
    Bo, _, Ho, Wo = ref.shape
    emb = None
    if use_emb:
        emb = prng.normal(seed, "cv.emb", (B, cout + 24))
        ref = ref + emb[:, 8:8 + cout, None, None]
    res = None
    if use_res:
        res = prng.normal(seed, "cv.res", (B, cout, Ho, Wo))
        ref = ref + res
    hi, lo = ops.pack_conv_weight(w.to(dev), prec)
    out = torch.full((B, Ho, Wo, cout), float("nan"), device=dev)
    ops.conv_igemm(nhwc(x1).to(dev), hi, lo, out, prec=prec, ks=ks, mode=m,
                   src2=None if x2 is None else nhwc(x2).to(dev), src2_bmod=bmod,
                   scale=None if sc is None else sc.to(dev), shift=None if sh is None else sh.to(dev), act=act,
                   bias=bias.to(dev), emb=None if emb is None else emb.to(dev), emb_offset=8,
                   emb_bstride=0 if emb is None else emb.shape[1], res=None if res is None else nhwc(res).to(dev))
    torch.cuda.synchronize()
    err = rel_err(nchw(out), ref)
    assert err < tol, f"{prec_name}: rel err {err:.3e} >= {tol}"
    return err


@pytest.mark.parametrize("prec,tol", PRECS)
@pytest.mark.parametrize("B,H,W,c1,c2,cout", [
    (2, 8, 8, 64, 0, 64),       # whole-image tiles, 2 samples per tile
    (3, 8, 8, 128, 64, 96),     # odd batch (partial tile), concat, cout < 128 (N mask)
    (2, 16, 16, 128, 0, 256),   # row tiles, 2 N tiles
    (1, 32, 32, 128, 0, 128),   # level-0 shape
    (5, 4, 4, 32, 32, 32),      # 4x4 images, 8 per tile, partial
    (1, 64, 64, 32, 0, 32),     # 2 rows per tile
])
def test_conv3x3_s1(dev, prec, tol, B, H, W, c1, c2, cout):
    _conv_case(dev, prec, tol, B, H, W, c1, c2, cout, "s1", 3, gn=True, act=1, use_emb=True, use_res=True)


@pytest.mark.parametrize("prec,tol", PRECS[:2])
def test_conv3x3_plain_and_bmod(dev, prec, tol):
    _conv_case(dev, prec, tol, 2, 16, 16, 64, 0, 64, "s1", 3, gn=False, act=0, use_emb=False, use_res=False)
    _conv_case(dev, prec, tol, 4, 8, 8, 64, 64, 128, "s1", 3, gn=True, act=1, use_emb=True, use_res=False, bmod=2)


@pytest.mark.parametrize("prec,tol", PRECS[:2])
@pytest.mark.parametrize("B,H,W,c,cout", [(2, 16, 16, 64, 64), (2, 32, 32, 128, 128), (3, 8, 8, 32, 32), (1, 64, 64, 32, 64)])
def test_conv_down(dev, prec, tol, B, H, W, c, cout):
    _conv_case(dev, prec, tol, B, H, W, c, 0, cout, "down", 3, gn=False, act=0, use_emb=False, use_res=False)


@pytest.mark.parametrize("prec,tol", PRECS[:2])
@pytest.mark.parametrize("B,H,W,c,cout", [(2, 8, 8, 64, 64), (2, 16, 16, 128, 128), (3, 4, 4, 32, 32), (1, 32, 32, 32, 32)])
def test_conv_up(dev, prec, tol, B, H, W, c, cout):
    _conv_case(dev, prec, tol, B, H, W, c, 0, cout, "up", 3, gn=False, act=0, use_emb=False, use_res=False)


@pytest.mark.parametrize("prec,tol", PRECS)
@pytest.mark.parametrize("B,H,W,c1,c2,cout,gn", [(2, 8, 8, 128, 64, 128, False), (3, 4, 4, 128, 0, 384, True), (1, 10, 10, 64, 0, 64, True)])
def test_conv1x1(dev, prec, tol, B, H, W, c1, c2, cout, gn):
    _conv_case(dev, prec, tol, B, H, W, c1, c2, cout, "s1", 1, gn=gn, act=0, use_emb=False, use_res=True)


def test_conv_rejects_bad_args(dev):
    from stedm_amd import _lib, ops
    prec = ops.Precision.parse("f16")
    x = torch.zeros((1, 8, 8, 48), device=dev)   # 48 channels: not a multiple of 32
    w = torch.zeros((32, 9, 48), dtype=torch.int16, device=dev)
    with pytest.raises(_lib.StedmHipError):
        ops.conv_igemm(x, w, None, torch.zeros((1, 8, 8, 32), device=dev), prec=prec)


# ------------------------------------------------------------------------------------------------ boundary convs
@pytest.mark.parametrize("B,H,W,c1,c2,cout", [(2, 32, 32, 4, 3, 128), (3, 16, 16, 7, 0, 32), (1, 128, 128, 3, 3, 128)])
def test_conv_in(dev, B, H, W, c1, c2, cout):
    from stedm_amd import ops
    x1 = prng.normal(3, "ci.x1", (B, c1, H, W))
    x2 = prng.normal(3, "ci.x2", (B, c2, H, W)) if c2 else None
    w = prng.normal(3, "ci.w", (cout, c1 + c2, 3, 3), 0.1)
    b = prng.normal(3, "ci.b", (cout,), 0.05)
    ref = F.conv2d(x1 if x2 is None else torch.cat([x1, x2], 1), w, b, padding=1)
    out = torch.empty((B, H, W, cout), device=dev)
    ops.conv_in(x1.to(dev), None if x2 is None else x2.to(dev), w.to(dev), b.to(dev), out)
    assert rel_err(nchw(out), ref) < 2e-5


@pytest.mark.parametrize("B,H,W,c,cout", [(2, 32, 32, 128, 4), (2, 16, 16, 32, 4), (1, 128, 128, 128, 3), (1, 40, 40, 64, 3)])
def test_conv_out(dev, B, H, W, c, cout):
    from stedm_amd import ops
    x = prng.normal(4, "co.x", (B, c, H, W)) * 1.4 + 0.2
    g = prng.normal(4, "co.g", (c,), 0.1, 1.0)
    bt = prng.normal(4, "co.bt", (c,), 0.1)
    w = prng.normal(4, "co.w", (cout, c, 3, 3), 0.03)
    b = prng.normal(4, "co.b", (cout,), 0.05)
    ref = F.conv2d(F.silu(F.group_norm(x, 32, g, bt, 1e-5)), w, b, padding=1)
    out = torch.empty((B, cout, H, W), device=dev)
    ops.conv_out(nhwc(x).to(dev), g.to(dev), bt.to(dev), 1e-5, 32, w.to(dev), b.to(dev), out)
    assert rel_err(out, ref) < 5e-5


# ------------------------------------------------------------------------------------------------ embeddings
@pytest.mark.parametrize("mc,nrow", [(128, 9), (32, 9), (128, 1), (128, 2), (32, 1)])
def test_time_embed_and_proj(dev, mc, nrow):
    """nrow <= 2 takes the few-row GEMV kernel (the uniform-timestep sampling path evaluates one row)."""
    from oracle import unet as ou
    from stedm_amd import ops
    ted = mc * 4
    t = torch.tensor([0, 1, 500, 999, 951, 21, 7, 333, 2], dtype=torch.long)[-nrow:]
    B = t.shape[0]
    w0 = prng.normal(5, "te.w0", (ted, mc), 1 / math.sqrt(mc)); b0 = prng.normal(5, "te.b0", (ted,), 0.05)
    w2 = prng.normal(5, "te.w2", (ted, ted), 1 / math.sqrt(ted)); b2 = prng.normal(5, "te.b2", (ted,), 0.05)
    ref = F.linear(F.silu(F.linear(ou.timestep_embedding(t, mc), w0, b0)), w2, b2)
    half = mc // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half).to(dev)
    emb = torch.empty((B, ted), device=dev)
    ops.time_embed(t.to(dev), freqs, ops.transpose(w0.to(dev)), b0.to(dev), ops.transpose(w2.to(dev)), b2.to(dev), emb)
    assert rel_err(emb, ref) < 1e-4
    ntot = 1000
    we = prng.normal(5, "ep.w", (ntot, ted), 1 / math.sqrt(ted)); be = prng.normal(5, "ep.b", (ntot,), 0.05)
    ref2 = F.linear(F.silu(ref), we, be)
    out = torch.empty((B, ntot), device=dev)
    ops.emb_proj(ref.to(dev), ops.transpose(we.to(dev)), be.to(dev), out)
    assert rel_err(out, ref2) < 2e-5


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("B,T,heads,ch", [(2, 64, 8, 128), (2, 16, 4, 32), (1, 256, 4, 32), (1, 100, 2, 64), (1, 1024, 8, 128)])
def test_attn_legacy(dev, B, T, heads, ch):
    from oracle import unet as ou
    from stedm_amd import ops
    qkv = prng.normal(6, "at.qkv", (B, heads * 3 * ch, T)) * 1.5   # reference layout [B, 3C, T]
    ref = ou.qkv_attention_legacy(qkv, heads)                       # [B, C, T]
    out = torch.empty((B, T, heads * ch), device=dev)
    ops.attn_legacy(qkv.permute(0, 2, 1).contiguous().to(dev), out, heads)
    assert rel_err(out.permute(0, 2, 1), ref) < 5e-5


@pytest.mark.parametrize("prec,tol", [("f16", 1e-2), ("bf16", 6e-2)])
@pytest.mark.parametrize("B,heads,ch", [(2, 8, 128), (3, 4, 64), (5, 2, 32), (130, 8, 128)])
def test_attn_legacy16_mfma(dev, prec, tol, B, heads, ch):
    """64-token attention on MFMA with 16-bit operands (single-product modes) against the fp32 oracle; output is the 16-bit plane."""
    from oracle import unet as ou
    from stedm_amd import ops
    pr = ops.Precision.parse(prec)
    T = 64
    qkv = prng.normal(7, "a16.qkv", (B, heads * 3 * ch, T)) * 1.5
    ref = ou.qkv_attention_legacy(qkv, heads)                       # [B, C, T]
    out = torch.empty((B, T, heads * ch), dtype=torch.int16, device=dev)
    qd = qkv.permute(0, 2, 1).contiguous().to(dev)
    ops.attn_legacy16(qd, out, heads, pr)
    got = _as_float(out, pr).permute(0, 2, 1)
    assert rel_err(got, ref) < tol
    # the same from a 16-bit qkv plane (what the qkv conv's epilogue writes): identical operand rounding, identical bits
    q16 = torch.empty(qd.shape, dtype=torch.int16, device=dev)
    ops.gn_apply16(qd.view(B, 1, T, -1), None, q16.view(B, 1, T, -1), None, pr)
    out2 = torch.empty_like(out)
    ops.attn_legacy16(q16, out2, heads, pr)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("prec,tol,tol16", [("f16", 2.5e-2, 1.2e-2), ("bf16", 2e-1, 8e-2)])      # (max error over more keys per query than at T = 64; reported modes)
@pytest.mark.parametrize("B,T,heads,ch", [(2, 256, 8, 128), (1, 1024, 8, 128), (3, 128, 4, 64), (66, 256, 8, 128),
                                          (2, 36, 8, 16), (1, 100, 2, 64), (2, 576, 8, 128), (3, 1000, 4, 32), (1, 4096, 2, 64), (5, 65, 3, 128), (2, 64, 8, 16)])
def test_attn_legacy16_mfma_key_tiles(dev, prec, tol, tol16, B, T, heads, ch):
    """the middle block's attention at larger latents (T = 256 at 64x64, 1024 at 128x128) and the SpatialTransformer's (any T, head widths
    16 .. 128): attn_flash_kernel - K / V rows of 64-key tiles staged once per workgroup, online softmax - from the 16-bit qkv plane,
    against the fp32 oracle (the mode's operand rounding included: tol) and against the same attention in fp64 on the ROUNDED operands
    (what the kernel itself adds: the 16-bit probabilities and outputs; tol16). Token counts off the 64-key / 128-query grid exercise the
    key mask and the dead query rows; ch = 16 the zero extension."""
    from oracle import unet as ou
    from stedm_amd import ops
    pr = ops.Precision.parse(prec)
    qkv = prng.normal(7, f"a16t.qkv.{T}", (B, heads * 3 * ch, T)) * 1.5
    ref = ou.qkv_attention_legacy(qkv, heads)                       # [B, C, T]
    qd = qkv.permute(0, 2, 1).contiguous().to(dev)
    q16 = torch.empty(qd.shape, dtype=torch.int16, device=dev)
    ops.gn_apply16(qd.view(B, 1, T, -1), None, q16.view(B, 1, T, -1), None, pr)
    out = torch.full((B, T, heads * ch), 0x7e7e, dtype=torch.int16, device=dev)
    ops.attn_legacy16(q16, out, heads, pr)
    got = _as_float(out, pr).permute(0, 2, 1)
    err = rel_err(got, ref)
    qr = _as_float(q16, pr).double().cpu().view(B, T, heads, 3, ch)
    q_, k_, v_ = qr[:, :, :, 0], qr[:, :, :, 1], qr[:, :, :, 2]      # [B, T, heads, ch]
    w = torch.softmax(torch.einsum("bthc,bshc->bhts", q_, k_) / math.sqrt(ch), dim=-1)
    ref16 = torch.einsum("bhts,bshc->bthc", w, v_).reshape(B, T, heads * ch).permute(0, 2, 1)
    err16 = rel_err(got, ref16)
    print(f"[attention T={T} B={B} {heads}x{ch} {prec}] vs fp32 oracle {err:.2e}, vs fp64 on the rounded operands {err16:.2e}")
    assert err < tol and err16 < tol16


@pytest.mark.parametrize("M,N,K,ta,tb", [(64, 512, 512, False, False), (64, 10368, 512, False, False), (10368, 512, 64, True, False), (64, 512, 10368, False, False),
                                         (512, 512, 64, True, False), (64, 512, 512, False, True), (225, 3, 512, False, False), (9, 512, 128, False, False),
                                         (100, 77, 45, True, True), (33, 130, 1500, False, True), (1, 64, 2048, False, False), (300, 20, 31, True, False)])
def test_gemm_f32_tiled_kernel_all_layouts(dev, M, N, K, ta, tb):
    """stedm_gemm_f32 (sgemm.hpp: 64x64 / 64x16 / 16x64 / 16x16 tiles, K step 32 with register prefetch, split K with a fixed-order reduce) in
    the four operand layouts, on the embedding path's shapes, on sizes that are multiples of nothing, with alpha / beta and through strided row
    views, against fp64 (fp32 products and sums: 2e-6 of the row scale); bitwise the same on a second run."""
    from stedm_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g).to(dev)
    Bm = torch.randn((N, K) if tb else (K, N), generator=g).to(dev)
    wide = torch.randn((M, N + 5), generator=g).to(dev)
    C0 = wide[:, 2:2 + N]                                  # a strided view (ldc = N + 5)
    ref = 0.5 * ((A.double().t() if ta else A.double()) @ (Bm.double().t() if tb else Bm.double())) + 0.25 * C0.double()
    ws = torch.empty((1 << 22,), device=dev)
    outs = []
    for rep in range(2):
        C = wide.clone()[:, 2:2 + N]
        ops.gemm_f32(A, ta, Bm, tb, C, alpha=0.5, beta=0.25, ws=ws)
        outs.append(C.clone())
    err = float((outs[0].double() - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
    assert err < 2e-6 * max(1.0, (K / 512) ** 0.5), err
    assert torch.equal(outs[0], outs[1])
    C = torch.full((M, N), float("nan"), device=dev)           # beta = 0 never reads the destination
    ops.gemm_f32(A, ta, Bm, tb, C)
    assert bool(torch.isfinite(C).all())


@pytest.mark.parametrize("B,K,N,act_in,act_out,bias", [(64, 512, 10368, 1, 0, True), (64, 128, 512, 0, 1, True), (64, 512, 512, 0, 0, True), (256, 768, 512, 2, 2, True),
                                                       (9, 100, 77, 1, 2, False), (225, 2, 512, 0, 2, True), (225, 512, 3, 0, 0, False), (17, 33, 4, 0, 0, True),
                                                       (2, 512, 10368, 1, 0, True), (8, 512, 512, 0, 1, True), (300, 512, 10368, 1, 0, True), (70, 3000, 516, 2, 1, True)])
def test_linear_kernel_forms_vs_fp64(dev, B, K, N, act_in, act_out, bias):
    """stedm_linear: act_out(bias + act_in(x) wt) with wt K-major. The weight-stream kernels (2 rows; blocks of 8 rows with x resident in LDS)
    while few row blocks re-read the weights, the tiled GEMM of sgemm.hpp (activations and bias in its load / store paths) for many rows x wide
    outputs, odd N or long K: the time_embed and emb_layers Linears at sampling / training batches, Agg_Linear, the Swin position-bias MLP.
    Against fp64 on the same activations."""
    from stedm_amd import ops
    g = torch.Generator().manual_seed(B + 3 * K + 7 * N)
    x = torch.randn((B, K), generator=g).to(dev)
    wt = (torch.randn((K, N), generator=g) / K ** 0.5).to(dev)
    b = torch.randn((N,), generator=g).to(dev) if bias else None
    act = lambda v, a: torch.nn.functional.silu(v) if a == 1 else (torch.relu(v) if a == 2 else v)
    ref = act(act(x.double(), act_in) @ wt.double() + (0 if b is None else b.double()), act_out)
    out = ops.linear(x, wt, b, torch.full((B, N), float("nan"), device=dev), act_in=act_in, act_out=act_out)
    err = float((out.double() - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
    assert err < 3e-6, err


def test_attn_flash_eight_wave_form_equals_the_four_wave_form(dev, tmp_path):
    """attn_flash_kernel has two workgroup forms: 4 waves / two-stage rings (128 queries) and 8 waves / four-stage rings (256 queries; chosen
    for T >= 1024 at 128-wide heads when the grid fills the chip, i.e. only at bench-sized batches). A wave does the same arithmetic in both, so
    the outputs must be bitwise equal: each form is forced through STEDM_ATTN_FORM in its own process (the switch is read once) on shapes that
    give the 8-wave form dead waves, a partial last key tile and every head width; the 4-wave results are the ones the tests above pin."""
    import subprocess
    import sys
    import os
    code = (
        "import sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "from stedm_amd import ops\n"
        "from stedm_amd.utils import prng\n"
        "dev = torch.device('cuda:0'); outs = {}\n"
        "for prec in ('f16', 'bf16'):\n"
        "    pr = ops.Precision.parse(prec)\n"
        "    for (B, T, H, ch) in [(2, 1100, 2, 128), (1, 1024, 3, 128), (3, 300, 2, 64), (2, 77, 4, 32), (2, 513, 2, 16)]:\n"
        "        qkv = (prng.normal(9, f'af.{T}.{ch}', (B, T, 3 * H * ch)) * 1.5).to(dev)\n"
        "        q16 = torch.empty(qkv.shape, dtype=torch.int16, device=dev)\n"
        "        ops.gn_apply16(qkv.view(B, 1, T, -1), None, q16.view(B, 1, T, -1), None, pr)\n"
        "        out = torch.full((B, T, H * ch), 0x7e7e, dtype=torch.int16, device=dev)\n"
        "        ops.attn_legacy16(q16, out, H, pr)\n"
        "        outs[f'{prec}.{T}.{ch}'] = out.cpu()\n"
        "torch.save(outs, sys.argv[1])\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for form in ("4", "8"):
        path = str(tmp_path / f"form{form}.pt")
        r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, STEDM_ATTN_FORM=form), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res[form] = torch.load(path, weights_only=True)
    assert set(res["4"]) == set(res["8"]) and len(res["4"]) == 10
    for k in res["4"]:
        assert torch.equal(res["4"][k], res["8"][k]), k
        assert bool((res["4"][k] != 0x7e7e).any())


# ------------------------------------------------------------------------------------------------ DDIM
@pytest.mark.parametrize("B,C,H,W,cfg,eta", [(2, 4, 32, 32, True, 0.0), (3, 3, 16, 16, True, 1.0), (2, 4, 8, 8, False, 1.0),
                                             (1, 3, 128, 128, True, 0.0), (2, 4, 12, 24, True, 0.5),
                                             (2, 4, 16, 16, True, 1.0), (3, 8, 8, 32, False, 0.5), (64, 4, 32, 32, True, 1.0)])   # (register-resident forms: 4 / 8 / 16 elements per thread)
def test_ddim_step(dev, B, C, H, W, cfg, eta):
    from oracle import ddim as od
    from stedm_amd import ops
    sched = od.Schedule()
    ds = od.DDIMSchedule(sched, 20, eta)
    x = prng.normal(7, "dd.x", (B, C, H, W))
    ec = prng.normal(7, "dd.ec", (B, C, H, W))
    eu = prng.normal(7, "dd.eu", (B, C, H, W)) * 0.8 + 0.1 * ec
    nz = prng.normal(7, "dd.nz", (B, C, H, W))
    idx = 10
    e = od.cfg_combine(ec, eu, 1.5) if cfg else ec
    xp_ref, x0_ref = od.ddim_update(x, e, *ds.scalars(idx), noise=nz)
    table = torch.tensor([ds.scalars(i) for i in range(20)], dtype=torch.float32, device=dev)
    step = torch.tensor([idx], dtype=torch.int32, device=dev)
    xp = torch.empty((B, C, H, W), device=dev)
    x0 = torch.empty((B, C, H, W), device=dev)
    ops.ddim_step(x.to(dev), ec.to(dev), eu.to(dev) if cfg else None, table, xp, pred_x0=x0, noise=nz.to(dev), step_idx=step,
                  cfg_scale=1.5, rescale_phi=0.7)
    assert rel_err(xp, xp_ref) < 2e-5
    assert rel_err(x0, x0_ref) < 2e-5
    ops.step_advance(step, -1)
    assert int(step.item()) == idx - 1


# ------------------------------------------------------------------------------------------------ two-kernel GN + DMA conv (v3)
def _as_float(t16, prec):
    from stedm_amd._lib import F16
    return t16.view(torch.float16 if prec.mm_dtype == F16 else torch.bfloat16).float()


@pytest.mark.parametrize("B,H,W,c1,c2,bmod", [(2, 8, 8, 64, 0, 0), (3, 16, 16, 128, 0, 0), (2, 8, 8, 1024, 512, 0),
                                              (4, 8, 8, 128, 128, 2), (2, 32, 32, 128, 0, 0), (2, 16, 16, 512, 128, 0)])
def test_gn_stats_apply16(dev, B, H, W, c1, c2, bmod):
    from stedm_amd import ops
    prec = ops.Precision.parse("parity")
    C = c1 + c2
    x1 = prng.normal(11, "g2.x1", (B, c1, H, W)) * 1.7 + 0.3
    x2 = prng.normal(11, "g2.x2", (bmod or B, c2, H, W)) * 0.6 - 0.2 if c2 else None
    g = prng.normal(11, "g2.g", (C,), 0.1, 1.0)
    b = prng.normal(11, "g2.b", (C,), 0.1)
    x2f = None if x2 is None else (x2 if not bmod else x2.repeat(B // bmod, 1, 1, 1))
    xin = x1 if x2 is None else torch.cat([x1, x2f], 1)
    ref = F.silu(F.group_norm(xin, 32, g, b, 1e-5))
    stats = torch.empty((B * ops.gn_nslab(C, H * W) * 32 * 2,), dtype=torch.float64, device=dev)
    d1 = nhwc(x1).to(dev); d2 = None if x2 is None else nhwc(x2).to(dev)
    ops.gn_stats(d1, d2, stats, 32, bmod)
    hi = torch.empty((B, H, W, C), dtype=torch.int16, device=dev); lo = torch.empty_like(hi)
    ops.gn_apply16(d1, d2, hi, lo, prec, g.to(dev), b.to(dev), 1e-5, 32, 1, stats, bmod)
    got = nchw(_as_float(hi, prec) + _as_float(lo, prec))
    assert rel_err(got, ref) < 2e-5
    assert rel_err(nchw(_as_float(hi, prec)), ref) < 1e-2   # hi plane alone: fp16 rounding of values up to ~8 std
    # plain conversion (no norm, no act)
    ops.gn_apply16(d1, d2, hi, lo, prec, x2_bmod=bmod)
    assert rel_err(nchw(_as_float(hi, prec) + _as_float(lo, prec)), xin) < 1e-6


@pytest.mark.parametrize("B,H,W,c1,c2,bmod", [(2, 8, 8, 64, 0, 0), (3, 16, 16, 128, 0, 0), (2, 8, 8, 1024, 512, 0), (4, 8, 8, 128, 128, 2),
                                              (2, 32, 32, 128, 0, 0), (2, 16, 16, 512, 128, 0), (2, 32, 32, 512, 128, 0), (4, 16, 16, 1024, 512, 2),
                                              (3, 4, 4, 1280, 0, 0), (2, 20, 20, 96, 32, 0),
                                              (200, 8, 8, 256, 256, 0), (400, 8, 8, 256, 0, 0), (200, 8, 8, 256, 256, 100), (128, 4, 4, 1024, 1024, 0)])
def test_gn_chan_stats_apply16c(dev, B, H, W, c1, c2, bmod):
    """GroupNorm from per-(sample, slab, channel) partials: group boundaries straddle the concat seam (e.g. 1024+512: cpg 48),
    ragged last slab (20x20), x2 shared modulo bmod; dual output (normalised + plain conversion planes). The large-batch 8 x 8 / 4 x 4
    cases take the channel-run form of the kernel (a block = a run of whole groups of every pixel of its sample)."""
    from stedm_amd import ops
    prec = ops.Precision.parse("parity")
    C = c1 + c2
    x1 = prng.normal(13, "g3.x1", (B, c1, H, W)) * 1.7 + 0.3
    x2 = prng.normal(13, "g3.x2", (bmod or B, c2, H, W)) * 0.6 - 0.2 if c2 else None
    g = prng.normal(13, "g3.g", (C,), 0.1, 1.0)
    b = prng.normal(13, "g3.b", (C,), 0.1)
    x2f = None if x2 is None else (x2 if not bmod else x2.repeat(B // bmod, 1, 1, 1))
    xin = x1 if x2 is None else torch.cat([x1, x2f], 1)
    ref = F.silu(F.group_norm(xin, 32, g, b, 1e-5))
    d1 = nhwc(x1).to(dev); d2 = None if x2 is None else nhwc(x2).to(dev)
    ns = ops.gn_chan_nslab(H * W)
    cs1 = torch.full((B, ns, c1, 2), float("nan"), device=dev)
    ops.gn_chan_stats(d1, cs1)
    # partials against a direct per-slab sum
    flat = d1.view(B, H * W, c1).cpu().double()
    for k in range(ns):
        sl = flat[:, k * 256:(k + 1) * 256]
        assert torch.allclose(cs1[:, k, :, 0].cpu().double(), sl.sum(1), rtol=1e-5, atol=1e-3)
        assert torch.allclose(cs1[:, k, :, 1].cpu().double(), (sl * sl).sum(1), rtol=1e-5, atol=1e-3)
    cs2 = None
    if d2 is not None:
        cs2 = torch.empty((d2.shape[0], ns, c2, 2), device=dev)
        ops.gn_chan_stats(d2, cs2)
    hi = torch.empty((B, H, W, C), dtype=torch.int16, device=dev); lo = torch.empty_like(hi)
    rhi = torch.empty_like(hi); rlo = torch.empty_like(hi)
    ops.gn_apply16c(d1, cs1, d2, cs2, hi, lo, prec, g.to(dev), b.to(dev), 1e-5, 32, 1, bmod, raw=(rhi, rlo))
    assert rel_err(nchw(_as_float(hi, prec) + _as_float(lo, prec)), ref) < 2e-5
    assert rel_err(nchw(_as_float(rhi, prec) + _as_float(rlo, prec)), xin) < 1e-6
    # bitwise reproducible; the {mean, rstd} side output (training: kept for the backward) equals the separate fold and F.group_norm's statistics
    hi2 = torch.empty_like(hi); lo2 = torch.empty_like(hi)
    mr = torch.full((B, 32, 2), float("nan"), device=dev)
    ops.gn_apply16c(d1, cs1, d2, cs2, hi2, lo2, prec, g.to(dev), b.to(dev), 1e-5, 32, 1, bmod, mean_rstd=mr)
    assert torch.equal(hi, hi2) and torch.equal(lo, lo2)
    if not bmod:
        mr2 = torch.empty_like(mr)
        ops.gn_fold(cs1, cs2, 32, H * W, 1e-5, mr2)
        assert torch.allclose(mr, mr2, rtol=1e-6, atol=1e-7)
    xg = xin.double().view(B, 32, -1)
    assert torch.allclose(mr[..., 0].cpu().double(), xg.mean(-1), rtol=1e-5, atol=1e-5)
    assert torch.allclose(mr[..., 1].cpu().double(), 1.0 / torch.sqrt(xg.var(-1, unbiased=False) + 1e-5), rtol=1e-4)


@pytest.mark.parametrize("precs", ["bf16", "f16", "bf16x3"])
@pytest.mark.parametrize("B,H,W,C", [(3, 32, 32, 128), (2, 16, 16, 1536), (5, 8, 8, 1024), (2, 20, 20, 96), (1, 4, 4, 32), (2, 64, 64, 4)])
def test_gn_chan_stats16_equals_stats_plus_cast(dev, precs, B, H, W, C):
    """The statistics pass that also leaves the 16-bit planes (training backward: a gradient's bias sums and its dgrad / wgrad operand from
    ONE read): the partial sums are those of gn_chan_stats (to rounding; bitwise run to run), the planes bitwise those of the plain conversion kernel, and the hi plane
    is torch's round-to-nearest-even cast; ragged last slot (20 x 20), narrow (4) and wide (1536) channel counts."""
    from stedm_amd import ops
    from stedm_amd._lib import BF16, F16
    prec = ops.Precision(F16 if precs == "f16" else BF16, 3 if precs.endswith("x3") else 1)
    x = (prng.normal(17, "cs16.x", (B, H, W, C)) * 2.3e-3 + 1e-4).to(dev)          # gradient-sized values
    ns = ops.gn_chan_nslab(H * W)
    cs_a = torch.full((B, ns, C, 2), float("nan"), device=dev); cs_b = torch.full_like(cs_a, float("nan"))
    hi = torch.full((B, H, W, C), -1, dtype=torch.int16, device=dev)
    lo = torch.full_like(hi, -1) if prec.npass == 3 else None
    ops.gn_chan_stats(x, cs_a)
    ops.gn_chan_stats16(x, cs_b, hi, lo, prec)
    # (the two instantiations contract a*a + q differently: equal to rounding, not bitwise; each is deterministic run to run)
    assert torch.allclose(cs_a, cs_b, rtol=1e-5, atol=1e-8)
    cs_c = torch.empty_like(cs_b); hi_c = torch.empty_like(hi)
    ops.gn_chan_stats16(x, cs_c, hi_c, None if lo is None else torch.empty_like(lo), prec)
    assert torch.equal(cs_b, cs_c) and torch.equal(hi, hi_c)
    rhi = torch.empty_like(hi); rlo = torch.empty_like(hi) if lo is not None else None
    ops.gn_apply16(x, None, rhi, rlo, prec)
    assert torch.equal(hi, rhi) and (lo is None or torch.equal(lo, rlo))
    tdt = torch.float16 if precs == "f16" else torch.bfloat16
    assert torch.equal(hi.view(tdt), x.to(tdt))
    if lo is not None:
        assert torch.equal(lo.view(tdt), (x - hi.view(tdt).float()).to(tdt))


@pytest.mark.parametrize("prec", ["f16", "bf16", "parity"])
@pytest.mark.parametrize("B,H,W,cin,cout", [(64, 32, 32, 32, 128), (200, 16, 16, 64, 96), (801, 8, 8, 32, 160), (2, 8, 8, 64, 64), (3, 32, 32, 64, 64)])
def test_conv_epilogue_chan_stats(dev, prec, B, H, W, cin, cout):
    """stedm_conv_args.chan_stats: statistics of the stored output (bias / emb / residual included), from the register-streamed
    kernel's epilogue (large cases, single product) or the dispatcher's extra pass (small cases, parity mode)."""
    from stedm_amd import ops
    pr = ops.Precision.parse(prec)
    x = torch.randn(B, H, W, cin, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) / math.sqrt(cin * 9)
    bias = torch.randn(cout, device=dev); emb = torch.randn(B, cout, device=dev); res = torch.randn(B, H, W, cout, device=dev)
    hi16 = torch.empty((B, H, W, cin), dtype=torch.int16, device=dev); lo16 = torch.empty_like(hi16)
    ops.gn_apply16(x, None, hi16, lo16, pr)
    whi, wlo = ops.pack_conv_weight(w, pr)
    ns = ops.gn_chan_nslab(H * W)
    out = torch.empty(B, H, W, cout, device=dev)
    cs = torch.full((B, ns, cout, 2), float("nan"), device=dev)
    ops.conv_igemm(None, whi, wlo, out, prec=pr, src16=(hi16, lo16), bias=bias, emb=emb, emb_bstride=cout, res=res,
                   w_frag=ops.pack_conv_weight_frag(w, pr) if pr.npass == 1 else None, chan_stats=cs)
    flat = out.view(B, H * W, cout).double()
    for k in range(ns):
        sl = flat[:, k * 256:(k + 1) * 256]
        assert torch.allclose(cs[:, k, :, 0].double(), sl.sum(1), rtol=1e-4, atol=2e-3)
        assert torch.allclose(cs[:, k, :, 1].double(), (sl * sl).sum(1), rtol=1e-4, atol=2e-3)
    cs2 = torch.full_like(cs, float("nan"))
    ops.conv_igemm(None, whi, wlo, out, prec=pr, src16=(hi16, lo16), bias=bias, emb=emb, emb_bstride=cout, res=res,
                   w_frag=ops.pack_conv_weight_frag(w, pr) if pr.npass == 1 else None, chan_stats=cs2)
    assert torch.equal(cs, cs2)   # no atomics anywhere: bitwise reproducible


@pytest.mark.parametrize("prec", ["f16", "bf16", "parity"])
@pytest.mark.parametrize("B,H,W,cin,cout", [(1, 24, 24, 512, 512), (2, 24, 24, 256, 128), (1, 40, 40, 512, 256), (3, 10, 10, 1024, 1024)])
def test_conv1x1_split_k_chan_stats_on_sizes_off_the_256_pixel_run(dev, prec, B, H, W, cin, cout):
    """a 1x1 (the attention block's proj_out: chan_stats + residual + workspace) whose H * W neither divides nor is a
    multiple of the 256-pixel statistics run, at a batch small enough for the K split (24 x 24 = 576 pixels, B = 1, 512 channels: 8 shares).
    The reduce pass cannot fill the caller's slots in that case and must not claim it did: the dispatcher's statistics pass writes them.
    Statistics are compared with the sums of the stored output; a stale buffer (the bug) leaves the NaN fill in place. (The U-Net classes
    themselves only admit latent widths the 3x3 tiles divide - powers of two - so this is reachable through the C ABI only.)"""
    from stedm_amd import ops
    pr = ops.Precision.parse(prec)
    x = torch.randn(B, H, W, cin, device=dev)
    w = torch.randn(cout, cin, 1, 1, device=dev) / math.sqrt(cin)
    bias = torch.randn(cout, device=dev); res = torch.randn(B, H, W, cout, device=dev)
    hi16 = torch.empty((B, H, W, cin), dtype=torch.int16, device=dev); lo16 = torch.empty_like(hi16)
    ops.gn_apply16(x, None, hi16, lo16, pr)
    whi, wlo = ops.pack_conv_weight(w, pr)
    ns = ops.gn_chan_nslab(H * W)
    out = torch.full((B, H, W, cout), float("nan"), device=dev)
    cs = torch.full((B, ns, cout, 2), float("nan"), device=dev)
    ops.conv_igemm(None, whi, wlo, out, prec=pr, ks=1, src16=(hi16, lo16), bias=bias, res=res,
                   w_frag=ops.pack_conv_weight_frag(w, pr) if pr.npass == 1 else None,
                   w_frag16=ops.pack_conv_weight_frag16(w, pr) if pr.npass == 3 else None,
                   chan_stats=cs, ws=torch.empty(16 * out.numel(), device=dev))
    xr = _as_float(hi16, pr).double() + (_as_float(lo16, pr).double() if pr.npass == 3 else 0.0)
    from stedm_amd._lib import F16
    wr = w.double() if pr.npass == 3 else w.to(torch.float16 if pr.mm_dtype == F16 else torch.bfloat16).double()
    ref = torch.einsum("bhwc,oc->bhwo", xr, wr.view(cout, cin)) + bias.double() + res.double()
    assert rel_err(out, ref) < (1e-4 if pr.npass == 3 else 2e-3)
    flat = out.view(B, H * W, cout).double()
    assert bool(torch.isfinite(cs).all()), "statistics slots left unwritten"
    for k in range(ns):
        sl = flat[:, k * 256:(k + 1) * 256]
        assert torch.allclose(cs[:, k, :, 0].double(), sl.sum(1), rtol=1e-4, atol=2e-3)
        assert torch.allclose(cs[:, k, :, 1].double(), (sl * sl).sum(1), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("prec", ["f16", "bf16"])
@pytest.mark.parametrize("B,H,W,c1,c2,bmod", [(128, 8, 8, 1024, 1024, 64), (128, 16, 16, 1024, 512, 64), (64, 32, 32, 512, 128, 0), (6, 16, 16, 512, 128, 3), (3, 8, 8, 1024, 512, 0),
                                              (2, 32, 32, 128, 128, 0)])
def test_gn_apply16c_x16_equals_the_fp32_source_pass_on_the_rounded_values(dev, prec, B, H, W, c1, c2, bmod):
    """stedm_gn_apply16c_x16 (round 5): the decoder concat's GroupNorm when the h half already sits in the raw plane as 16-bit values (no fp32
    tensor of it exists). Against stedm_gn_apply16c fed the SAME rounded values as fp32 and the same channel statistics: bitwise the same
    normalised planes; the raw plane keeps the h half and receives the plain conversion of the skip half. Straddling groups (1536, 640
    channels), CFG batch sharing (bmod), the channel-cut block form of small samples."""
    from stedm_amd import ops
    pr = ops.Precision.parse(prec)
    tdt = torch.float16 if prec == "f16" else torch.bfloat16
    C = c1 + c2
    x1 = torch.randn(B, H, W, c1, device=dev) * 1.7 + 0.2
    B2 = bmod if bmod else B
    x2 = torch.randn(B2, H, W, c2, device=dev) * 0.8 - 0.1
    gamma = torch.randn(C, device=dev) * 0.3 + 1.0; beta = torch.randn(C, device=dev) * 0.2
    cs1 = torch.empty((B, ops.gn_chan_nslab(H * W), c1, 2), device=dev); ops.gn_chan_stats(x1, cs1)
    cs2 = torch.empty((B2, ops.gn_chan_nslab(H * W), c2, 2), device=dev); ops.gn_chan_stats(x2, cs2)
    x1r = x1.to(tdt)                                                 # what the producing convolution's epilogue stores
    raw = torch.full((B, H, W, C), 0x7e7e, dtype=torch.int16, device=dev)
    raw.view(tdt)[..., :c1] = x1r
    out = torch.full((B, H, W, C), 0x7e7e, dtype=torch.int16, device=dev)
    ops.gn_apply16c_x16(c1, cs1, x2, cs2, out, raw, pr, gamma, beta, 1e-5, 32, 1, bmod)
    ref_out = torch.empty_like(out); ref_raw = torch.empty_like(raw)
    ops.gn_apply16c(x1r.float().contiguous(), cs1, x2, cs2, ref_out, None, pr, gamma, beta, 1e-5, 32, 1, bmod, (ref_raw, None))
    assert torch.equal(out, ref_out)
    assert torch.equal(raw, ref_raw)
    assert torch.equal(raw.view(tdt)[..., :c1], x1r)


@pytest.mark.parametrize("prec", ["f16", "bf16"])
@pytest.mark.parametrize("B,H,W,cin,cout,extra,mode,res,ws", [
    (128, 16, 16, 512, 512, 512, "s1", True, False),      # a decoder ResBlock tail at the 16 x 16 level (identity-skip form)
    (2, 8, 8, 1024, 1024, 1024, "s1", True, True),        # K split: the reduce pass writes the 16-bit values and the statistics
    (128, 8, 8, 1024, 1024, 1024, "s1", False, True),     # the headline step's 8 x 8 level (two-way K split)
    (128, 8, 8, 512, 256, 64, "up2", False, False),       # Upsample into the next concat's plane (sub-pixel form, scattered rows)
    (3, 8, 8, 256, 128, 128, "up2", False, True)])
def test_conv_16bit_only_output_with_statistics_into_a_wider_plane(dev, prec, B, H, W, cin, cout, extra, mode, res, ws):
    """stedm_conv_args.out16_stride (round 5): out == NULL, the 16-bit output goes into channels [0, cout) of a [.., cout + extra] plane, the
    statistics are still written. Against the same launch with an fp32 output and a contiguous out16 plane: the same 16-bit values and
    the same statistics, bit for bit; the other channels of the wide plane untouched."""
    from stedm_amd import ops
    from stedm_amd._lib import CONV_S1, CONV_UP_SUBPIXEL
    pr = ops.Precision.parse(prec)
    up = mode == "up2"
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    x = torch.randn(B, H, W, cin, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) / math.sqrt(cin * 9)
    bias = torch.randn(cout, device=dev)
    r = torch.randn(B, Ho, Wo, cout, device=dev) if res else None
    h16 = torch.empty((B, H, W, cin), dtype=torch.int16, device=dev)
    ops.gn_apply16(x, None, h16, None, pr)
    if up:
        whi, wlo = ops.pack_conv_weight_up(w, pr); wf = ops.pack_conv_weight_up_frag(w, pr)
        ns = 4 * ops.gn_chan_nslab(H * W)
    else:
        whi, wlo = ops.pack_conv_weight(w, pr); wf = ops.pack_conv_weight_frag(w, pr)
        ns = ops.gn_chan_nslab(H * W)
    kw = dict(prec=pr, mode=CONV_UP_SUBPIXEL if up else CONV_S1, src16=(h16, None), bias=bias, res=r, w_frag=wf)
    mk_ws = lambda: torch.empty(16 * B * Ho * Wo * cout, device=dev) if ws else None
    out = torch.empty(B, Ho, Wo, cout, device=dev); o16 = torch.zeros((B, Ho, Wo, cout), dtype=torch.int16, device=dev)
    cs = torch.full((B, ns, cout, 2), float("nan"), device=dev)
    ops.conv_igemm(None, whi, wlo, out, chan_stats=cs, out16=(o16, None), ws=mk_ws(), **kw)
    wide = torch.full((B, Ho, Wo, cout + extra), 0x7e7e, dtype=torch.int16, device=dev)
    cs2 = torch.full_like(cs, float("nan"))
    kwc = dict(kw, chan_stats=cs2, out16=(wide, None), out16_stride=cout + extra, cout=cout, ws=mk_ws())
    assert ops.conv_igemm(None, whi, wlo, None, query_rs=True, **kwc)
    ops.conv_igemm(None, whi, wlo, None, **kwc)
    assert torch.equal(wide[..., :cout], o16)
    assert bool((wide[..., cout:] == 0x7e7e).all())
    assert torch.equal(cs, cs2)


@pytest.mark.parametrize("prec,B,H,W,m16", [("f16", 64, 32, 32, False), ("bf16", 3, 32, 32, False), ("f16", 128, 32, 32, False), ("f16", 5, 16, 32, False),
                                          ("bf16", 128, 16, 32, False), ("bf16", 7, 32, 24, False), ("bf16", 67, 32, 32, False)])
def test_conv_group_norm_epilogue_across_the_tiles_of_a_sample(dev, prec, B, H, W, m16):
    """stedm_conv_args.gn_coop (round 5): where a sample spans 2 .. 4 tiles of 256 pixels (32 x 32 and 16 x 32 pixels at 128 channels) the tiles
    exchange their channel sums INSIDE the launch (8-byte {epoch, value} words, write-through, polled by one wave) and each normalises its own
    rows: the GroupNorm + SiLU planes of ResBlock.out_layers come out of the convolution's epilogue and no fp32 tensor is stored. Against the
    same convolution with an fp32 output followed by the stedm_gn_apply16c pass (same statistics, folded in the same order): the planes agree
    to one unit of the 16-bit format, over eight forwards that reuse the word block under advancing epochs, with a second stream keeping
    part of the chip busy (uneven arrival of a sample's tiles) and with the word block poisoned by a stale epoch; the give-up flag stays 0.
    A 24-wide grid (the generic-shape form of the convolution) ends with the separate pass as before."""
    from stedm_amd import ops
    pr = ops.Precision.parse(prec)
    c = 128
    g = torch.Generator().manual_seed(B * 131 + H * 7 + W)
    x = torch.randn(B, H, W, c, generator=g).to(dev)
    w = (torch.randn(c, c, 3, 3, generator=g) / math.sqrt(c * 9)).to(dev)
    bias = torch.randn(c, generator=g).to(dev)
    emb = torch.randn(B, c, generator=g).to(dev)
    gamma, beta = (1.0 + 0.3 * torch.randn(c, generator=g)).to(dev), (0.2 * torch.randn(c, generator=g)).to(dev)
    h16 = torch.empty((B, H, W, c), dtype=torch.int16, device=dev)
    ops.gn_apply16(x, None, h16, None, pr)
    whi, wlo = ops.pack_conv_weight(w, pr); wf = ops.pack_conv_weight_frag(w, pr)
    kw = dict(prec=pr, src16=(h16, None), bias=bias, emb=emb, emb_offset=0, emb_bstride=c, w_frag=wf)
    ns = ops.gn_chan_nslab(H * W)
    # reference: fp32 output + the separate pass
    out = torch.empty(B, H, W, c, device=dev); cs = torch.empty(B, ns, c, 2, device=dev)
    ops.conv_igemm(None, whi, wlo, out, chan_stats=cs, **kw)
    ref = torch.empty((B, H, W, c), dtype=torch.int16, device=dev)
    ops.gn_apply16c(out, cs, None, None, ref, None, pr, gamma, beta, 1e-5, 32, 1)
    asf = lambda t: t.view(torch.float16 if prec == "f16" else torch.bfloat16).float()
    unit = 2.0 ** (-10 if prec == "f16" else -7)
    words = torch.zeros((B, 4, 128, 2), dtype=torch.int64, device=dev)
    cw = ops.coop_words_new()
    side = torch.cuda.Stream()
    big = torch.randn(4096, 4096, device=dev)
    for it in range(8):
        ops.step_advance(cw, 1)
        if it == 5:      # words of some other forward: tags that are not this epoch must be ignored like the zeros of a fresh block
            words.fill_((int(cw[0].item()) + 1000) << 32 | 0x3f800000)
        if it >= 3:      # uneven arrival: another stream holds part of the chip
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    big @ big
        o2 = torch.full((B, H, W, c), float("nan"), device=dev); cs2 = torch.empty_like(cs)
        planes = torch.full((B, H, W, c), 0x7e7e, dtype=torch.int16, device=dev)
        ops.conv_igemm(None, whi, wlo, o2, chan_stats=cs2, gn_next=(gamma, beta, 1e-5, 32, 1, planes, None, True), coop=(words, cw), **kw)
        torch.cuda.synchronize()
        assert torch.equal(cs, cs2)
        d = (asf(planes) - asf(ref)).abs()
        lim = unit * asf(ref).abs().clamp_min(2.0 ** -14) * 1.01
        assert bool((d <= lim).all()), (it, float(d.max()))
        assert float((planes != ref).float().mean()) < 2e-3, "more than rounding-boundary cases differ"
        if W == 32 and B >= 64:      # (a grid that fills the chip: smaller ones take other kernels / split K and end with the pass)
            assert bool(torch.isnan(o2).all()), "the epilogue form did not run (it stores no fp32 tensor: gn_only)"
    assert int(cw[1].item()) == 0
    ops.coop_check("test")
    if B == 64:      # the host side of a give-up: the flag a tile sets when its bounded spin ends raises at the next check and is cleared by it
        from stedm_amd._lib import StedmHipError
        cw[1:2].fill_(1)
        with pytest.raises(StedmHipError, match="gave up waiting"):
            ops.f16_guard_check("test")
        ops.coop_check("test")


def _conv_dma_case(dev, prec_name, tol, B, Hin, Win, cin, cout, mode, ks, use_emb=True, use_res=True, seed=12, frag=False, ws=False, m16=False, want_rs=None):
    from stedm_amd import ops
    from stedm_amd._lib import CONV_DOWN, CONV_S1, CONV_UP, CONV_UP_SUBPIXEL
    prec = ops.Precision.parse(prec_name)
    x = prng.normal(seed, "cd.x", (B, cin, Hin, Win))
    a = F.silu(x * 1.3 + 0.1)                      # stands for the normalised + activated activation
    w = prng.normal(seed, "cd.w", (cout, cin, ks, ks), 1.0 / math.sqrt(cin * ks * ks))
    bias = prng.normal(seed, "cd.b", (cout,), 0.05)
    if mode == "s1":
        ref = F.conv2d(a, w, bias, padding=ks // 2); m = CONV_S1
    elif mode == "down":
        ref = F.conv2d(a, w, bias, stride=2, padding=1); m = CONV_DOWN
    else:
        ref = F.conv2d(F.interpolate(a, scale_factor=2, mode="nearest"), w, bias, padding=1)
        m = CONV_UP_SUBPIXEL if mode == "up2" else CONV_UP
    _, _, Ho, Wo = ref.shape
    emb = res = None
    if use_emb:
        emb = prng.normal(seed, "cd.emb", (B, cout + 24)); ref = ref + emb[:, 8:8 + cout, None, None]
    if use_res:
        res = prng.normal(seed, "cd.res", (B, cout, Ho, Wo)); ref = ref + res
    hi16 = torch.empty((B, Hin, Win, cin), dtype=torch.int16, device=dev); lo16 = torch.empty_like(hi16)
    ops.gn_apply16(nhwc(a).to(dev), None, hi16, lo16, prec)
    whi, wlo = ops.pack_conv_weight_up(w.to(dev), prec) if mode == "up2" else ops.pack_conv_weight(w.to(dev), prec)
    out = torch.full((B, Ho, Wo, cout), float("nan"), device=dev)
    # stride-2 patches can exceed LDS in the DMA kernel: pass the fp32 source too so the dispatcher may fall back
    src1 = nhwc(a).to(dev) if mode == "down" else None
    kw = dict(prec=prec, ks=ks, mode=m, src16=(hi16, lo16), bias=bias.to(dev),
              emb=None if emb is None else emb.to(dev), emb_offset=8, emb_bstride=0 if emb is None else emb.shape[1],
              res=None if res is None else nhwc(res).to(dev),
              w_frag=None if not frag else (ops.pack_conv_weight_up_frag(w.to(dev), prec) if mode == "up2" else ops.pack_conv_weight_frag(w.to(dev), prec)),
              ws=torch.empty(16 * out.numel(), device=dev) if ws else None,
              w_frag16=None if not m16 else (ops.pack_conv_weight_up_frag16_hl(w.to(dev), prec) if mode == "up2"
                                             else ops.pack_conv_weight_frag16(w.to(dev), prec)))
    if want_rs is not None:     # the register-streamed kernel must (not) be the one that runs
        assert ops.conv_igemm(src1, whi, wlo, out, query_rs=True, **kw) == want_rs
    ops.conv_igemm(src1, whi, wlo, out, **kw)
    torch.cuda.synchronize()
    err = rel_err(nchw(out), ref)
    assert err < tol, f"{prec_name}: rel err {err:.3e} >= {tol}"


@pytest.mark.parametrize("prec,tol", PRECS)
@pytest.mark.parametrize("B,H,W,cin,cout", [
    (2, 8, 8, 64, 64), (5, 8, 8, 192, 96), (2, 16, 16, 128, 256), (1, 32, 32, 128, 128), (9, 4, 4, 64, 32),
    (1, 64, 64, 32, 32), (64, 8, 8, 128, 128), (8, 32, 32, 64, 128), (1, 128, 128, 32, 32)])
def test_conv_dma_3x3(dev, prec, tol, B, H, W, cin, cout):
    _conv_dma_case(dev, prec, tol, B, H, W, cin, cout, "s1", 3)


@pytest.mark.parametrize("prec", ["parity", "parity_bf16"])
@pytest.mark.parametrize("B,H,W,cin,cout,emb,res,ws", [
    (128, 8, 8, 2048, 1024, False, False, False),     # a decoder skip_connection of the NS32 step
    (128, 16, 16, 640, 512, True, True, False), (200, 16, 16, 64, 160, False, True, False), (801, 8, 8, 96, 32, True, False, False),
    (3, 128, 128, 64, 64, False, True, False), (130, 8, 8, 1024, 3072, False, False, False),     # an AttentionBlock's qkv
    (2, 16, 16, 512, 512, False, True, True), (4, 8, 8, 1024, 256, True, True, True)])          # small grids: K split over the workspace
def test_conv_1x1_three_product_register_streamed(dev, prec, B, H, W, cin, cout, emb, res, ws):
    """1x1 convolutions of the 3-product modes (skip_connection openaimodel.py:254, qkv / proj_out :343-346) on the register-streamed
    kernel (conv_rs.inc RS_1X1M: 16x16x32 MFMA, hi + lo activation planes and fragment streams): against fp64 at the split-product
    tolerance; ragged sample counts, cout not a multiple of 128, K split."""
    tol = {"parity": 2e-5, "parity_bf16": 3e-4}[prec]        # max error over the output's spread: 22- / 16-bit operand products, K up to 2048
    _conv_dma_case(dev, prec, tol, B, H, W, cin, cout, "s1", 1, use_emb=emb, use_res=res, ws=ws, m16=True, want_rs=True)


@pytest.mark.parametrize("prec,tol", PRECS[1:])
@pytest.mark.parametrize("B,H,W,cin,cout,emb,res", [
    (50, 32, 32, 32, 96, True, True), (200, 16, 16, 64, 128, True, False), (801, 8, 8, 32, 32, False, True), (3, 128, 128, 32, 64, True, True),
    (12, 64, 64, 32, 32, False, False), (64, 32, 32, 64, 160, True, True), (1601, 4, 4, 64, 128, True, True)])
def test_conv_dma_3x3_frag_weights(dev, prec, tol, B, H, W, cin, cout, emb, res):
    """256-row tile kernel with register-streamed fragment-order weights (conv_rs.inc); shapes fill >= 192 tiles so it is
    the kernel the dispatcher picks; ragged sample counts, cout not a multiple of 128, partial last tile."""
    _conv_dma_case(dev, prec, tol, B, H, W, cin, cout, "s1", 3, use_emb=emb, use_res=res, frag=True)


@pytest.mark.parametrize("prec,tol", PRECS[1:])
@pytest.mark.parametrize("B,H,W,cin,cout,emb,res,ws", [
    (50, 32, 32, 256, 96, True, True, False), (200, 16, 16, 288, 128, True, False, False), (801, 8, 8, 256, 32, False, True, False),
    (3, 128, 128, 256, 64, True, True, False), (12, 64, 64, 320, 32, False, False, False), (64, 32, 32, 256, 160, True, True, False),
    (1601, 4, 4, 256, 128, True, True, False), (128, 8, 8, 1024, 1024, True, True, False), (64, 16, 16, 1536, 512, False, False, False),
    (64, 8, 8, 2048, 1024, True, False, True), (2, 8, 8, 1024, 1024, False, True, True), (1, 32, 32, 256, 128, True, True, True),
    (64, 32, 32, 128, 128, True, True, False)])
def test_conv_3x3_mfma16x16x32_kind(dev, prec, tol, B, H, W, cin, cout, emb, res, ws):
    """the 3x3 kind on v_mfma_f32_16x16x32 (conv_rs.inc RS_3X3M: two skewed 16-channel planes per 32-channel chunk, uniform tap
    offsets, fragment-order weights of stedm_pack_conv_weight_frag16): full grids, ragged batches, partial N tiles, the bench's
    large-K shapes, and the split-K form of small grids — against F.conv2d. (Below 256 input channels the dispatcher keeps the
    32x32x16 form — the last case — which is faster there.)"""
    _conv_dma_case(dev, prec, tol, B, H, W, cin, cout, "s1", 3, use_emb=emb, use_res=res, frag=True, ws=ws, m16=True)


@pytest.mark.parametrize("prec", ["f16", "bf16"])
@pytest.mark.parametrize("B,H,W,cin,cout,ws", [
    (128, 32, 32, 256, 128, False),     # 512 tiles of rows inside one sample: two tile rounds per CU
    (128, 16, 16, 256, 512, False),     # 512 tiles over 4 N-tiles: the XCD-aware order
    (201, 8, 8, 256, 256, False),       # whole-sample tiles (4 samples each), ragged batch: masked rows, one slot per sample
    (1601, 4, 4, 256, 128, False),      # 16 samples per tile, one row fragment each
    (50, 32, 32, 256, 96, False),       # partial N tile: masked columns
    (96, 32, 32, 288, 128, False),      # 384 tiles on 256 CUs; odd chunk count
    (3, 128, 128, 256, 64, False),      # 128-pixel rows, two rows per tile
    (4, 8, 8, 1024, 1024, True),        # small grid: K split into the workspace, statistics from the reduce pass
    (64, 8, 8, 2048, 1024, True)])
def test_conv_3x3_16x16x32_full_epilogue(dev, prec, B, H, W, cin, cout, ws):
    """the 16x16x32 3x3 kind with everything its epilogue carries at once — bias, per-sample embedding row (offset + batch stride), residual
    rows, per-channel statistics in 256-pixel slots — over the grid shapes of the bench (two tile rounds per CU, the XCD-aware order, whole-sample
    tiles with a ragged batch, partial N tiles, 384 tiles on 256 CUs, split K): output against F.conv2d at the mode's tolerance, statistics
    against sums of the stored output, and bitwise run to run."""
    from stedm_amd import ops
    pr = ops.Precision.parse(prec)
    tol = dict(PRECS)[prec]
    g = torch.Generator(device="cpu").manual_seed(B * 7 + cin)
    a = F.silu(torch.randn(B, cin, H, W, generator=g) * 1.3 + 0.1)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    bias = torch.randn(cout, generator=g) * 0.05
    emb = torch.randn(B, cout + 24, generator=g)
    res = torch.randn(B, cout, H, W, generator=g)
    ref = F.conv2d(a, w, bias, padding=1) + emb[:, 8:8 + cout, None, None] + res
    hi16 = torch.empty((B, H, W, cin), dtype=torch.int16, device=dev)
    ops.gn_apply16(nhwc(a).to(dev), None, hi16, None, pr)
    whi, wlo = ops.pack_conv_weight(w.to(dev), pr)
    ns = ops.gn_chan_nslab(H * W)
    kw = dict(prec=pr, ks=3, src16=(hi16, None), bias=bias.to(dev), emb=emb.to(dev), emb_offset=8, emb_bstride=emb.shape[1], res=nhwc(res).to(dev),
              w_frag=ops.pack_conv_weight_frag(w.to(dev), pr), w_frag16=ops.pack_conv_weight_frag16(w.to(dev), pr))
    outs, css = [], []
    for _ in range(2):
        out = torch.full((B, H, W, cout), float("nan"), device=dev)
        cs = torch.full((B, ns, cout, 2), float("nan"), device=dev)
        ops.conv_igemm(None, whi, wlo, out, chan_stats=cs, ws=torch.empty(16 * out.numel(), device=dev) if ws else None, **kw)
        outs.append(out); css.append(cs)
    torch.cuda.synchronize()
    err = rel_err(nchw(outs[0]), ref)
    assert err < tol, f"{prec}: rel err {err:.3e} >= {tol}"
    flat = outs[0].view(B, H * W, cout).double()
    for k in range(ns):
        sl = flat[:, k * 256:(k + 1) * 256]
        assert torch.allclose(css[0][:, k, :, 0].double(), sl.sum(1), rtol=1e-4, atol=2e-3)
        assert torch.allclose(css[0][:, k, :, 1].double(), (sl * sl).sum(1), rtol=1e-4, atol=2e-3)
    assert torch.equal(outs[0], outs[1]) and torch.equal(css[0], css[1])          # no atomics, fixed summation orders


@pytest.mark.parametrize("B,H,W,cin,cout,emb,res,ws", [
    (50, 32, 32, 256, 96, True, True, False), (200, 16, 16, 288, 128, True, False, False), (801, 8, 8, 256, 32, False, True, False),
    (12, 64, 64, 128, 32, True, True, False), (64, 32, 32, 128, 160, True, True, False), (400, 8, 8, 320, 256, True, True, False),
    (128, 8, 8, 1024, 1024, True, True, False), (64, 16, 16, 1536, 512, False, False, False), (64, 8, 8, 2048, 1024, True, False, True),
    (2, 8, 8, 1024, 1024, False, True, True), (1, 32, 32, 384, 128, True, True, True), (100, 16, 16, 160, 256, True, True, False)])
def test_conv_3x3_three_products_register_streamed(dev, B, H, W, cin, cout, emb, res, ws):
    """the 3-product (hi / lo split) mode on the register-streamed 16x16x32 kind (conv_rs.inc P3: hi + lo images in a 3-buffer chunk
    ring, hi + lo fragment streams of stedm_pack_conv_weight_frag16_hl, lo.hi + hi.lo + hi.hi per fragment pair): full and ragged
    grids, whole-sample tiles, odd and even chunk counts, the split-K form — against F.conv2d at the parity tolerance; the capability
    query confirms which kernel runs."""
    _conv_dma_case(dev, "parity", PRECS[0][1], B, H, W, cin, cout, "s1", 3, use_emb=emb, use_res=res, ws=ws, m16=True, want_rs=True)


@pytest.mark.parametrize("prec,tol", PRECS[1:])
@pytest.mark.parametrize("B,H,W,cin,cout,res", [(128, 8, 8, 2048, 1024, False), (200, 16, 16, 192, 96, True), (64, 32, 32, 64, 128, False),
                                                (801, 8, 8, 128, 160, True), (50, 32, 32, 640, 128, True), (13, 64, 64, 64, 32, False)])
def test_conv_dma_1x1_frag_weights(dev, prec, tol, B, H, W, cin, cout, res):
    """1x1 convolution as 4 register-streamed 16-channel slices per barrier (conv_rs_kernel<.., 1, 4>)."""
    _conv_dma_case(dev, prec, tol, B, H, W, cin, cout, "s1", 1, use_emb=False, use_res=res, frag=True)


@pytest.mark.parametrize("prec,tol", PRECS[1:])
@pytest.mark.parametrize("B,H,W,cin,cout,res", [(2, 8, 8, 2048, 1024, False), (1, 16, 16, 1536, 512, True), (4, 8, 8, 1024, 3072, False), (3, 32, 32, 640, 128, True)])
def test_conv_dma_1x1_split_k(dev, prec, tol, B, H, W, cin, cout, res):
    """small-batch 1x1 (skip_connection, qkv): K split over several blocks per tile with the workspace."""
    _conv_dma_case(dev, prec, tol, B, H, W, cin, cout, "s1", 1, use_emb=False, use_res=res, frag=True, ws=True)


@pytest.mark.parametrize("prec,tol", PRECS[1:])
@pytest.mark.parametrize("B,H,W,c,cout", [(128, 8, 8, 64, 128), (50, 16, 16, 128, 96), (201, 8, 8, 32, 32), (13, 32, 32, 64, 160), (803, 4, 4, 32, 128)])
def test_conv_dma_up_subpixel_frag_weights(dev, prec, tol, B, H, W, c, cout):
    """sub-pixel upsample through the register-streamed kernel (conv_rs_kernel<.., 4, 2>, 4 parities x tiles)."""
    _conv_dma_case(dev, prec, tol, B, H, W, c, cout, "up2", 3, use_emb=True, use_res=True, frag=True)


@pytest.mark.parametrize("prec,tol", PRECS[1:])
@pytest.mark.parametrize("m16", [False, True])
@pytest.mark.parametrize("B,H,W,cin,cb,cout,emb", [(64, 32, 32, 32, 64, 128, False), (200, 16, 16, 64, 192, 96, True), (801, 8, 8, 32, 128, 160, False),
                                                   (128, 8, 8, 512, 1024, 768, False), (50, 32, 32, 128, 640, 128, True), (13, 64, 64, 32, 64, 32, False),
                                                   (128, 16, 16, 512, 1536, 512, True), (128, 32, 32, 256, 640, 128, True), (201, 8, 8, 1024, 2048, 1024, False)])
def test_conv_fused_skip(dev, prec, tol, B, H, W, cin, cb, cout, emb, m16):
    """conv3x3(h) + conv1x1(x) + both biases in one kernel (ResBlock tail `skip_connection(x) + h`, openaimodel.py:288):
    the 1x1 runs as a second K-phase of the register-streamed 3x3 kernel; per-channel statistics of the sum come with it.
    m16: with the 16x16x32-MFMA fragment orders at hand (both phases on that shape from 256 input channels on)."""
    from stedm_amd import ops
    pr = ops.Precision.parse(prec)
    hsrc = F.silu(prng.normal(21, "fs.h", (B, cin, H, W)))
    x = prng.normal(21, "fs.x", (B, cb, H, W))
    w3 = prng.normal(21, "fs.w3", (cout, cin, 3, 3), 1.0 / math.sqrt(cin * 9))
    w1 = prng.normal(21, "fs.w1", (cout, cb, 1, 1), 1.0 / math.sqrt(cb))
    b3 = prng.normal(21, "fs.b3", (cout,), 0.05); b1 = prng.normal(21, "fs.b1", (cout,), 0.05)
    ref = F.conv2d(hsrc, w3, b3, padding=1) + F.conv2d(x, w1, b1)
    e = None
    if emb:
        e = prng.normal(21, "fs.e", (B, cout)); ref = ref + e[:, :, None, None]
    h16 = torch.empty((B, H, W, cin), dtype=torch.int16, device=dev); x16 = torch.empty((B, H, W, cb), dtype=torch.int16, device=dev)
    ops.gn_apply16(nhwc(hsrc).to(dev), None, h16, None, pr)
    ops.gn_apply16(nhwc(x).to(dev), None, x16, None, pr)
    whi, wlo = ops.pack_conv_weight(w3.to(dev), pr)
    out = torch.full((B, H, W, cout), float("nan"), device=dev)
    cs = torch.full((B, ops.gn_chan_nslab(H * W), cout, 2), float("nan"), device=dev)
    kw = dict(prec=pr, src16=(h16, None), bias=b3.to(dev), emb=None if e is None else e.to(dev), emb_bstride=0 if e is None else cout,
              w_frag=ops.pack_conv_weight_frag(w3.to(dev), pr), chan_stats=cs,
              w_frag16=ops.pack_conv_weight_frag16(w3.to(dev), pr) if m16 else None,
              skip=(x16, ops.pack_conv_weight_frag(w1.to(dev), pr), b1.to(dev)) + ((ops.pack_conv_weight_frag16(w1.to(dev), pr),) if m16 else ()))
    assert ops.conv_igemm(None, whi, wlo, out, query_fused=True, **kw)
    ops.conv_igemm(None, whi, wlo, out, **kw)
    torch.cuda.synchronize()
    err = rel_err(nchw(out), ref)
    assert err < tol, f"{prec}: rel err {err:.3e} >= {tol}"
    flat = out.view(B, H * W, cout).double()
    for k in range(cs.shape[1]):
        sl = flat[:, k * 256:(k + 1) * 256]
        assert torch.allclose(cs[:, k, :, 0].double(), sl.sum(1), rtol=1e-4, atol=2e-3)
    if cout % 32 == 0:
        # round 5: the NEXT block's in_layers GroupNorm + SiLU riding on this call (gn_next; in the epilogue where a tile holds whole samples and
        # whole groups - also behind the fused phase now - by the trailing pass otherwise): same fp32 output and statistics, planes equal to
        # the separate stedm_gn_apply16c pass to one unit of the 16-bit format
        gamma = (1.0 + 0.3 * prng.normal(21, "fs.g", (cout,))).to(dev); beta = (0.2 * prng.normal(21, "fs.bt", (cout,))).to(dev)
        ref16 = torch.empty((B, H, W, cout), dtype=torch.int16, device=dev)
        ops.gn_apply16c(out, cs, None, None, ref16, None, pr, gamma, beta, 1e-5, 32, 1)
        out2 = torch.full_like(out, float("nan")); cs2 = torch.full_like(cs, float("nan"))
        planes = torch.full((B, H, W, cout), 0x7e7e, dtype=torch.int16, device=dev)
        ops.conv_igemm(None, whi, wlo, out2, gn_next=(gamma, beta, 1e-5, 32, 1, planes, None, False), **dict(kw, chan_stats=cs2))
        assert torch.equal(out2, out) and torch.equal(cs2, cs)
        asf = lambda t: t.view(torch.float16 if pr.label == "f16" else torch.bfloat16).float()
        unit = 2.0 ** (-10 if pr.label == "f16" else -7)
        d = (asf(planes) - asf(ref16)).abs()
        assert bool((d <= unit * asf(ref16).abs().clamp_min(2.0 ** -14) * 1.01).all()), float(d.max())


@pytest.mark.parametrize("prec,tol", PRECS[1:])
@pytest.mark.parametrize("m16", [False, True])
@pytest.mark.parametrize("B,H,W,cin,cb,cout,emb", [(2, 8, 8, 1024, 2048, 1024, True), (2, 16, 16, 512, 1536, 512, False), (16, 8, 8, 1024, 2048, 1024, True),
                                                   (2, 32, 32, 128, 640, 128, True), (4, 16, 16, 512, 640, 512, False), (3, 8, 8, 512, 1024, 768, True)])
def test_conv_fused_skip_with_k_split(dev, prec, tol, B, H, W, cin, cb, cout, emb, m16):
    """The fused ResBlock tail on grids that leave CUs idle (sampling batches of 1 .. 8): every K share of the 3x3 also takes its share of the
    skip_connection's 64-channel chunks; the reduce pass adds both biases, the embedding row and emits the statistics. Against fp64; the
    second run is the first bit for bit."""
    from stedm_amd import ops
    pr = ops.Precision.parse(prec)
    hsrc = F.silu(prng.normal(23, "fsk.h", (B, cin, H, W)))
    x = prng.normal(23, "fsk.x", (B, cb, H, W))
    w3 = prng.normal(23, "fsk.w3", (cout, cin, 3, 3), 1.0 / math.sqrt(cin * 9))
    w1 = prng.normal(23, "fsk.w1", (cout, cb, 1, 1), 1.0 / math.sqrt(cb))
    b3 = prng.normal(23, "fsk.b3", (cout,), 0.05); b1 = prng.normal(23, "fsk.b1", (cout,), 0.05)
    ref = F.conv2d(hsrc.double(), w3.double(), b3.double(), padding=1) + F.conv2d(x.double(), w1.double(), b1.double())
    e = None
    if emb:
        e = prng.normal(23, "fsk.e", (B, cout)); ref = ref + e.double()[:, :, None, None]
    h16 = torch.empty((B, H, W, cin), dtype=torch.int16, device=dev); x16 = torch.empty((B, H, W, cb), dtype=torch.int16, device=dev)
    ops.gn_apply16(nhwc(hsrc).to(dev), None, h16, None, pr)
    ops.gn_apply16(nhwc(x).to(dev), None, x16, None, pr)
    whi, wlo = ops.pack_conv_weight(w3.to(dev), pr)
    runs = []
    for _ in range(2):
        out = torch.full((B, H, W, cout), float("nan"), device=dev)
        cs = torch.full((B, ops.gn_chan_nslab(H * W), cout, 2), float("nan"), device=dev)
        kw = dict(prec=pr, src16=(h16, None), bias=b3.to(dev), emb=None if e is None else e.to(dev), emb_bstride=0 if e is None else cout,
                  w_frag=ops.pack_conv_weight_frag(w3.to(dev), pr), chan_stats=cs, ws=torch.empty(16 * out.numel(), device=dev),
                  w_frag16=ops.pack_conv_weight_frag16(w3.to(dev), pr) if m16 else None,
                  skip=(x16, ops.pack_conv_weight_frag(w1.to(dev), pr), b1.to(dev)) + ((ops.pack_conv_weight_frag16(w1.to(dev), pr),) if m16 else ()))
        assert ops.conv_igemm(None, whi, wlo, out, query_fused=True, **kw)
        ops.conv_igemm(None, whi, wlo, out, **kw)
        runs.append((out, cs))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    out, cs = runs[0]
    err = rel_err(nchw(out), ref)
    assert err < tol, f"{prec}: rel err {err:.3e} >= {tol}"
    flat = out.view(B, H * W, cout).double()
    for k in range(cs.shape[1]):
        assert torch.allclose(cs[:, k, :, 0].double(), flat[:, k * 256:(k + 1) * 256].sum(1), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("prec,tol", PRECS[1:])
@pytest.mark.parametrize("B,H,W,cin,cout,mode,m16", [(8, 16, 512, 128, 128, "s1", False), (3, 17, 1024, 64, 96, "s1", False), (4, 24, 512, 256, 128, "s1", True),
                                                       (6, 32, 256, 256, 256, "s1", True), (4, 16, 128, 256, 128, "up2", False)])
def test_conv_rows_wider_than_the_tile(dev, prec, tol, B, H, W, cin, cout, mode, m16):
    """image rows of 256 .. 1024 pixels (the first stage's decoder at 256^2 / 512^2): a 256-pixel tile is a run of ONE row, its patch
    3 x 258 positions (conv_geometry wsplit) — 3x3 and the sub-pixel upsample on the register-streamed kernel, against F.conv2d."""
    _conv_dma_case(dev, prec, tol, B, H, W, cin, cout, mode, 3, use_emb=(mode == "s1"), use_res=True, frag=True, m16=m16)


def test_conv_fused_skip_rejected_when_unsupported(dev):
    """a fusion the kernel cannot run is reported by the query and refused by the call (never silently dropped)."""
    from stedm_amd import ops
    from stedm_amd._lib import StedmHipError
    pr = ops.Precision.parse("bf16")
    B, H, W, cin, cb, cout = 2, 8, 8, 32, 64, 32      # too few tiles for the 256-row kernel
    h16 = torch.zeros((B, H, W, cin), dtype=torch.int16, device=dev); x16 = torch.zeros((B, H, W, cb), dtype=torch.int16, device=dev)
    w3 = torch.randn(cout, cin, 3, 3, device=dev); w1 = torch.randn(cout, cb, 1, 1, device=dev)
    whi, wlo = ops.pack_conv_weight(w3, pr)
    out = torch.empty(B, H, W, cout, device=dev)
    kw = dict(prec=pr, src16=(h16, None), w_frag=ops.pack_conv_weight_frag(w3, pr), skip=(x16, ops.pack_conv_weight_frag(w1, pr), None))
    assert not ops.conv_igemm(None, whi, wlo, out, query_fused=True, **kw)
    with pytest.raises(StedmHipError):
        ops.conv_igemm(None, whi, wlo, out, **kw)


@pytest.mark.parametrize("prec,tol", PRECS[1:])
@pytest.mark.parametrize("B,H,W,cin,cout,emb,res", [(64, 8, 8, 1024, 1024, True, True), (100, 16, 16, 256, 128, False, True), (30, 32, 32, 256, 96, True, False),
                                                    (4, 8, 8, 1024, 1024, True, True), (2, 16, 16, 512, 512, False, True), (1, 32, 32, 256, 128, True, False),
                                                    (3, 8, 8, 512, 160, True, True)])
def test_conv_dma_3x3_split_k(dev, prec, tol, B, H, W, cin, cout, emb, res):
    """grids that leave CUs idle, with a workspace: K split over 2..16 blocks per tile (the small cases: 16 ways), partial tiles
    summed in a fixed order by the reduce kernel, which also applies bias / emb / residual and emits the channel statistics
    (bitwise reproducible)."""
    from stedm_amd import ops
    _conv_dma_case(dev, prec, tol, B, H, W, cin, cout, "s1", 3, use_emb=emb, use_res=res, frag=True, ws=True)
    pr = ops.Precision.parse(prec)
    x = torch.randn(B, H, W, cin, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) / math.sqrt(cin * 9)
    h16 = torch.empty((B, H, W, cin), dtype=torch.int16, device=dev)
    ops.gn_apply16(x, None, h16, None, pr)
    whi, wlo = ops.pack_conv_weight(w, pr); wf = ops.pack_conv_weight_frag(w, pr)
    outs = []
    for _ in range(3):
        out = torch.full((B, H, W, cout), float("nan"), device=dev)
        cs = torch.full((B, ops.gn_chan_nslab(H * W), cout, 2), float("nan"), device=dev)
        ops.conv_igemm(None, whi, wlo, out, prec=pr, src16=(h16, None), w_frag=wf, chan_stats=cs, ws=torch.empty(16 * out.numel(), device=dev))
        outs.append((out, cs))
    assert all(torch.equal(outs[0][0], o) and torch.equal(outs[0][1], c) for o, c in outs[1:])
    # 16-bit side output (operand planes of an Upsample consumer) comes out of the reduce kernel
    o16 = torch.zeros((B, H, W, cout), dtype=torch.int16, device=dev)
    out2 = torch.empty_like(outs[0][0])
    ops.conv_igemm(None, whi, wlo, out2, prec=pr, src16=(h16, None), w_frag=wf, ws=torch.empty(16 * out2.numel(), device=dev), out16=(o16, None))
    assert torch.equal(out2, outs[0][0])
    assert torch.equal(_as_float(o16, pr), out2.to(torch.float16 if prec == "f16" else torch.bfloat16).float())
    assert torch.allclose(outs[0][1][:, 0, :, 0].double(), outs[0][0].view(B, H * W, cout)[:, :256].double().sum(1), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("B,H,W,cin,cout,run", [(2, 8, 8, 1024, 1024, 8), (2, 16, 16, 512, 512, 32), (1, 32, 32, 256, 128, 64), (4, 16, 16, 512, 256, 128)])
def test_conv_split_k_caller_chosen_statistics_partition(dev, B, H, W, cin, cout, run):
    """A split-K convolution fills the statistics partition the caller laid out (equal runs of 8..256 pixels per slot, so that the
    reduce pass of a small batch has blocks for the whole chip): every slot holds its run's {sum x, sum x^2}, the output is the one
    of the default 256-pixel partition bit for bit, and a partition the kernel cannot serve leaves the statistics to the caller."""
    from stedm_amd import ops
    pr = ops.Precision.parse("bf16")
    x = torch.randn(B, H, W, cin, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) / math.sqrt(cin * 9)
    h16 = torch.empty((B, H, W, cin), dtype=torch.int16, device=dev)
    ops.gn_apply16(x, None, h16, None, pr)
    whi, wlo = ops.pack_conv_weight(w, pr); wf = ops.pack_conv_weight_frag(w, pr)
    HW = H * W
    ref = torch.empty(B, H, W, cout, device=dev)
    cs0 = torch.empty((B, ops.gn_chan_nslab(HW), cout, 2), device=dev)
    ops.conv_igemm(None, whi, wlo, ref, prec=pr, src16=(h16, None), w_frag=wf, chan_stats=cs0, ws=torch.empty(16 * ref.numel(), device=dev))
    out = torch.full_like(ref, float("nan"))
    cs = torch.full((B, HW // run, cout, 2), float("nan"), device=dev)
    ops.conv_igemm(None, whi, wlo, out, prec=pr, src16=(h16, None), w_frag=wf, chan_stats=cs, ws=torch.empty(16 * out.numel(), device=dev))
    assert torch.equal(out, ref)
    flat = out.view(B, HW // run, run, cout).double()
    assert torch.allclose(cs[..., 0].double(), flat.sum(2), rtol=1e-4, atol=2e-3)
    assert torch.allclose(cs[..., 1].double(), (flat * flat).sum(2), rtol=1e-4, atol=2e-3)
    assert torch.allclose(cs.sum(1).double(), cs0.sum(1).double(), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("prec", ["bf16", "f16"])
@pytest.mark.parametrize("B,H,W,cin,cout,groups,emb,res", [
    (2, 8, 8, 1024, 1024, 32, True, False),      # batch-1 CFG step: split 16, one group per block of the reduce pass
    (64, 8, 8, 512, 1024, 32, True, False),      # batch-64 encoder: split 2
    (2, 16, 16, 512, 512, 32, True, True),       # 256 pixels per sample, two groups per block
    (3, 16, 16, 256, 128, 32, False, False),     # 4 channels per group: eight groups per block
    (2, 8, 8, 256, 96, 8, True, False),          # 12 channels per group: 32 % 12 != 0 -> the follow-up pass
    (4, 32, 32, 128, 128, 32, True, False),      # 1024 pixels per sample -> the follow-up pass
    (128, 16, 16, 512, 512, 32, True, False),    # full grid, no split: one sample per tile, eight groups per 128 channels -> the conv epilogue
    (128, 8, 8, 1024, 1024, 32, True, True),     # four samples per tile, four groups per 128 channels -> the conv epilogue
    (126, 8, 8, 256, 128, 32, True, False),      # ... with a last tile of two samples and two masked ones
    (64, 16, 16, 128, 512, 32, True, False),     # the 32x32x16 MFMA kind (128 input channels)
    (96, 16, 16, 256, 320, 32, False, False),    # cout % 128 != 0 -> the follow-up pass
])
def test_conv_with_the_consumers_groupnorm(dev, prec, B, H, W, cin, cout, groups, emb, res):
    """stedm_conv_args.gn_*: the GroupNorm + SiLU that reads a convolution's output, written as 16-bit planes by the same call — inside the
    split-K reduce pass or the convolution's own epilogue where a workgroup owns whole groups of whole samples, by the stedm_gn_apply16c pass
    otherwise. Against the two calls made separately:
    identical fp32 output, statistics equal up to rounding; the planes equal up to one unit of the 16-bit format on a few elements (the fused
    pass adds the channel and group sums in another order), and within 1e-2 of the fp32 GroupNorm of the output everywhere."""
    from stedm_amd import ops
    pr = ops.Precision.parse(prec)
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + cout)
    x = torch.randn(B, H, W, cin, generator=g).to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).to(dev)
    bias = torch.randn(cout, generator=g).to(dev)
    embt = torch.randn(B, cout, generator=g).to(dev) if emb else None
    rest = torch.randn(B, H, W, cout, generator=g).to(dev) if res else None
    gamma = (1 + 0.2 * torch.randn(cout, generator=g)).to(dev); beta = (0.2 * torch.randn(cout, generator=g)).to(dev)
    h16 = torch.empty((B, H, W, cin), dtype=torch.int16, device=dev)
    ops.gn_apply16(x, None, h16, None, pr)
    whi, wlo = ops.pack_conv_weight(w, pr); wf = ops.pack_conv_weight_frag(w, pr)
    wf16 = ops.pack_conv_weight_frag16(w, pr) if cin >= 256 else None
    kw = dict(prec=pr, src16=(h16, None), bias=bias, emb=embt, emb_bstride=cout if emb else 0, res=rest, w_frag=wf, w_frag16=wf16)
    ns = ops.gn_chan_nslab(H * W)
    out_a = torch.empty(B, H, W, cout, device=dev); cs_a = torch.empty(B, ns, cout, 2, device=dev)
    ops.conv_igemm(None, whi, wlo, out_a, chan_stats=cs_a, ws=torch.empty(16 * out_a.numel(), device=dev), **kw)
    pl_a = torch.zeros(B, H, W, cout, dtype=torch.int16, device=dev)
    ops.gn_apply16c(out_a, cs_a, None, None, pl_a, None, pr, gamma, beta, 1e-5, groups, 1)
    out_b = torch.empty_like(out_a); cs_b = torch.empty_like(cs_a); pl_b = torch.zeros_like(pl_a)
    ops.conv_igemm(None, whi, wlo, out_b, chan_stats=cs_b, ws=torch.empty(16 * out_b.numel(), device=dev), gn_next=(gamma, beta, 1e-5, groups, 1, pl_b), **kw)
    # (the channel sums are added in the order of the pass's own pixel-lane layout: equal up to fp32 rounding, not bit for bit)
    assert torch.equal(out_a, out_b) and torch.allclose(cs_a, cs_b, rtol=2e-5, atol=1e-3)
    fa, fb = _as_float(pl_a, pr), _as_float(pl_b, pr)
    ulp = (2.0 ** -7 if prec == "bf16" else 2.0 ** -10)
    diff = (fa - fb).abs()
    assert float((diff > 0).float().mean()) < 1e-3 and bool((diff <= ulp * fa.abs().clamp_min(2.0 ** -14) * 1.01).all()), (float(diff.max()), float((diff > 0).float().mean()))
    ref = torch.nn.functional.silu(torch.nn.functional.group_norm(out_a.permute(0, 3, 1, 2), groups, gamma, beta, 1e-5)).permute(0, 2, 3, 1)
    assert float((fb - ref).abs().max()) < (2e-2 if prec == "bf16" else 4e-3) * max(1.0, float(ref.abs().max()))
    # gn_only: the planes are the output's only consumer — a launch whose epilogue writes them itself leaves `out` alone (all of it), any
    # other form still fills it; the planes are the same bits either way
    out_c = torch.full_like(out_a, float("nan")); cs_c = torch.empty_like(cs_a); pl_c = torch.zeros_like(pl_a)
    ops.conv_igemm(None, whi, wlo, out_c, chan_stats=cs_c, ws=torch.empty(16 * out_c.numel(), device=dev),
                   gn_next=(gamma, beta, 1e-5, groups, 1, pl_c, None, True), **kw)
    assert torch.equal(pl_c, pl_b) and torch.equal(cs_c, cs_b)
    untouched = bool(torch.isnan(out_c).all())
    assert untouched or torch.equal(out_c, out_a)
    if (B, H, cout) in ((128, 16, 512), (128, 8, 1024)):
        assert untouched, "the whole-sample tiles of this shape were expected to skip the fp32 store"
    if cin == cout:    # the planes an epilogue may write while other tiles still gather must not be the planes the convolution reads
        with pytest.raises(AssertionError):
            ops.conv_igemm(None, whi, wlo, out_c, chan_stats=cs_c, ws=torch.empty(16 * out_c.numel(), device=dev),
                           gn_next=(gamma, beta, 1e-5, groups, 1, h16), **kw)


@pytest.mark.parametrize("prec,tol", PRECS[1:])
@pytest.mark.parametrize("B,H,W,cin,cout,ws", [(64, 32, 32, 128, 128, True), (64, 16, 16, 512, 512, True), (3, 8, 8, 32, 32, False), (5, 16, 16, 64, 96, True),
                                                (2, 64, 64, 32, 160, False), (9, 8, 8, 64, 64, True)])
def test_conv_s2d_downsample(dev, prec, tol, B, H, W, cin, cout, ws):
    """Downsample.op (3x3 stride 2 pad 1, openaimodel.py:156-173) as a 2x2 stride-1 conv over space-to-depth planes on the
    register-streamed kernel; with a workspace the small grids split K. Statistics of the output come from the epilogue / reduce."""
    from stedm_amd import ops
    from stedm_amd._lib import CONV_S2D
    pr = ops.Precision.parse(prec)
    x = prng.normal(31, "sd.x", (B, cin, H, W))
    w = prng.normal(31, "sd.w", (cout, cin, 3, 3), 1.0 / math.sqrt(cin * 9))
    bias = prng.normal(31, "sd.b", (cout,), 0.05)
    ref = F.conv2d(x, w, bias, stride=2, padding=1)
    planes = torch.empty((B, H // 2, W // 2, 4 * cin), dtype=torch.int16, device=dev)
    ops.space_to_depth16(nhwc(x).to(dev), planes, None, pr)
    # the planes hold pixel (2y+py, 2x+px) in channel block py*2+px
    got = _as_float(planes, pr).view(B, H // 2, W // 2, 2, 2, cin).permute(0, 5, 1, 3, 2, 4).reshape(B, cin, H, W).cpu()
    assert rel_err(got, x) < (2e-2 if prec == "bf16" else 2e-3)   # 16-bit rounding of values up to ~4 sigma
    out = torch.full((B, H // 2, W // 2, cout), float("nan"), device=dev)
    cs = torch.full((B, ops.gn_chan_nslab(H * W // 4), cout, 2), float("nan"), device=dev)
    ops.conv_igemm(None, None, None, out, prec=pr, mode=CONV_S2D, src16=(planes, None), bias=bias.to(dev),
                   w_frag=ops.pack_conv_weight_s2d_frag(w.to(dev), pr), chan_stats=cs, ws=torch.empty(2 * out.numel(), device=dev) if ws else None)
    torch.cuda.synchronize()
    err = rel_err(nchw(out), ref)
    assert err < tol, f"{prec}: rel err {err:.3e} >= {tol}"
    flat = out.view(B, -1, cout).double()
    for k in range(cs.shape[1]):
        assert torch.allclose(cs[:, k, :, 0].double(), flat[:, k * 256:(k + 1) * 256].sum(1), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("prec,tol", PRECS[1:])
@pytest.mark.parametrize("B,H,W,cin,cout,res", [(2, 8, 8, 1024, 1024, False), (2, 16, 16, 512, 512, True), (16, 8, 8, 1024, 1024, False), (3, 16, 16, 256, 96, True),
                                                (1, 32, 32, 128, 128, False)])
def test_conv_subpixel_upsample_split_k(dev, prec, tol, B, H, W, cin, cout, res):
    """The sub-pixel Upsample on grids that leave CUs idle (a sampling batch of 1 .. 8: 4 parities x a few tiles): K split over the workspace
    on the register-streamed kernel — partial planes of the full-resolution output, summed in a fixed order by the reduce pass, which also
    fills the caller's statistics partition (4 slots per 256 low-resolution pixels); bitwise reproducible, against fp64."""
    from stedm_amd import ops
    from stedm_amd._lib import CONV_UP_SUBPIXEL
    pr = ops.Precision.parse(prec)
    _conv_dma_case(dev, prec, tol, B, H, W, cin, cout, "up2", 3, use_emb=False, use_res=res, frag=True, ws=True, want_rs=True)
    x = torch.randn(B, H, W, cin, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) / math.sqrt(cin * 9)
    h16 = torch.empty((B, H, W, cin), dtype=torch.int16, device=dev)
    ops.gn_apply16(x, None, h16, None, pr)
    whi, wlo = ops.pack_conv_weight_up(w, pr); wf = ops.pack_conv_weight_up_frag(w, pr)
    runs = []
    for _ in range(2):
        out = torch.full((B, 2 * H, 2 * W, cout), float("nan"), device=dev)
        cs = torch.full((B, 4 * ops.gn_chan_nslab(H * W), cout, 2), float("nan"), device=dev)
        ops.conv_igemm(None, whi, wlo, out, prec=pr, mode=CONV_UP_SUBPIXEL, src16=(h16, None), w_frag=wf, chan_stats=cs, ws=torch.empty(16 * out.numel(), device=dev))
        runs.append((out, cs))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    out, cs = runs[0]
    assert bool(torch.isfinite(out).all()) and bool(torch.isfinite(cs).all())
    flat = out.view(B, -1, cout).double()
    assert torch.allclose(cs[..., 0].sum(1).double(), flat.sum(1), rtol=1e-4, atol=2e-3)
    assert torch.allclose(cs[..., 1].sum(1).double(), (flat * flat).sum(1), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("prec", ["parity", "parity_bf16", "f16", "bf16"])
@pytest.mark.parametrize("B,H,W,cin,cout,emb,res,ws", [(128, 8, 8, 1024, 1024, False, False, False), (128, 16, 16, 512, 512, False, False, False),     # the NS32 step's two Upsamples
                                                       (50, 16, 16, 128, 96, True, True, False), (7, 32, 32, 160, 200, False, True, False), (130, 8, 8, 256, 64, True, False, False),
                                                       (2, 8, 8, 1024, 1024, False, False, True), (3, 16, 16, 256, 96, False, True, True)])      # small grids: K split
def test_conv_subpixel_upsample_16x16x32_register_streamed(dev, prec, B, H, W, cin, cout, emb, res, ws):
    """Upsample (nearest x2 + 3x3, openaimodel.py:122-132) on the register-streamed kernel's 16x16x32 form (conv_rs.inc RS_SUBM: four
    output parities with pre-summed 2x2 taps; 3-product modes: hi + lo planes and fragment streams; single-product modes, round 5: the hi
    stream of the same pack - the kind the headline step's two Upsamples run) against fp64; ragged batches / cout; K split (single product)."""
    if ws and prec.startswith("parity"):
        pytest.skip("the 3-product sub-pixel form has no K split")
    ws = ws or not prec.startswith("parity")      # (single product: a grid under 3/4 of the chip runs on this kernel only with the workspace)
    tol = {"parity": 2e-5, "parity_bf16": 3e-4, "f16": 5e-3, "bf16": 3e-2}[prec]
    _conv_dma_case(dev, prec, tol, B, H, W, cin, cout, "up2", 3, use_emb=emb, use_res=res, m16=True, want_rs=True, ws=ws)


@pytest.mark.parametrize("prec", ["parity", "parity_bf16", "f16", "bf16"])
@pytest.mark.parametrize("B,H,W,cin,cout,ws", [(128, 32, 32, 128, 128, False), (128, 16, 16, 512, 512, False), (64, 16, 16, 512, 512, True), (5, 16, 16, 64, 96, True),
                                                (40, 64, 64, 32, 160, False), (9, 8, 8, 64, 64, True)])
def test_conv_s2d_downsample_three_product(dev, prec, B, H, W, cin, cout, ws):
    """Downsample.op in the 3-product modes: hi + lo space-to-depth planes, the RS_SUBM kind with the space-to-depth filter's hi + lo
    fragment streams; K split over the workspace on small grids; statistics from the epilogue / reduce pass."""
    from stedm_amd import ops
    from stedm_amd._lib import CONV_S2D
    pr = ops.Precision.parse(prec)
    tol = {"parity": 2e-5, "parity_bf16": 3e-4, "f16": 5e-3, "bf16": 3e-2}[prec]      # (f16 / bf16, round 5: the hi stream of the same pack, RS_SUBM single product)
    x = prng.normal(33, "sd3.x", (B, cin, H, W))
    w = prng.normal(33, "sd3.w", (cout, cin, 3, 3), 1.0 / math.sqrt(cin * 9))
    bias = prng.normal(33, "sd3.b", (cout,), 0.05)
    ref = F.conv2d(x.double(), w.double(), bias.double(), stride=2, padding=1)
    hi = torch.empty((B, H // 2, W // 2, 4 * cin), dtype=torch.int16, device=dev); lo = torch.empty_like(hi)
    ops.space_to_depth16(nhwc(x).to(dev), hi, lo, pr)
    out = torch.full((B, H // 2, W // 2, cout), float("nan"), device=dev)
    cs = torch.full((B, ops.gn_chan_nslab(H * W // 4), cout, 2), float("nan"), device=dev)
    kw = dict(prec=pr, mode=CONV_S2D, src16=(hi, lo), bias=bias.to(dev), w_frag16=ops.pack_conv_weight_s2d_frag16_hl(w.to(dev), pr),
              ws=torch.empty(2 * out.numel(), device=dev) if ws else None)
    if not ops.conv_igemm(None, None, None, out, query_rs=True, **kw):
        # a problem the register-streamed kernel declines (a single tile with too short a K walk per share): what UNetModel then runs is the
        # fused fp32-source kernel (stedm_amd/unet.py, Downsample) — checked here the same way
        from stedm_amd._lib import CONV_DOWN
        whi, wlo = ops.pack_conv_weight(w.to(dev), pr)
        ops.conv_igemm(nhwc(x).to(dev), whi, wlo, out, prec=pr, mode=CONV_DOWN, bias=bias.to(dev))
        assert rel_err(nchw(out), ref) < tol
        return
    ops.conv_igemm(None, None, None, out, chan_stats=cs, **kw)
    torch.cuda.synchronize()
    err = rel_err(nchw(out), ref)
    assert err < tol, f"{prec}: rel err {err:.3e} >= {tol}"
    flat = out.view(B, -1, cout).double()
    for k in range(cs.shape[1]):
        assert torch.allclose(cs[:, k, :, 0].double(), flat[:, k * 256:(k + 1) * 256].sum(1), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("prec", ["bf16", "f16", "parity"])
@pytest.mark.parametrize("rows,dim", [(4098 * 2 + 3, 256), (1031, 512), (517, 768), (259, 1024), (66, 2048), (131, 320), (37, 64)])
def test_ln_apply16(dev, prec, rows, dim):
    """LayerNorm -> operand planes (vit_set.py:14-20 PreNorm, attention.py BasicTransformerBlock norms): the one-pass register form for
    dim = 256 k (rows per wave 4 / 2 / 1, ragged last wave) and the scalar form for any other width, against torch's layer_norm."""
    from stedm_amd import ops
    pr = ops.Precision.parse(prec)
    x = (torch.randn(rows, dim, device=dev) * 1.7 + 0.3)
    g = torch.randn(dim, device=dev) * 0.2 + 1.0; b = torch.randn(dim, device=dev) * 0.1
    hi = torch.full((rows, dim), 0x7FFF, dtype=torch.int16, device=dev)
    lo = torch.full((rows, dim), 0x7FFF, dtype=torch.int16, device=dev) if pr.npass == 3 else None
    ops.ln_apply16(x, g, b, 1e-5, hi, lo, pr)
    ref = F.layer_norm(x.double(), (dim,), g.double(), b.double(), 1e-5)
    got = _as_float(hi, pr).double() + (_as_float(lo, pr).double() if lo is not None else 0.0)
    assert rel_err(got, ref) < (2e-6 if pr.npass == 3 else (2e-2 if prec == "bf16" else 2.5e-3))     # max error / spread: half an ulp of a 6-sigma value


def test_chan_stats_any_partition(dev):
    """The statistics slots may cut a sample's pixels any way (256-pixel runs, row pairs from conv_in, run x parity from the
    sub-pixel upsample): producers with different partitions feed one GroupNorm over their concat."""
    from stedm_amd import ops
    prec = ops.Precision.parse("parity")
    B, H, W, c1, c2 = 3, 16, 16, 64, 32
    x1 = prng.normal(41, "ap.x1", (B, c1, H, W)) * 1.3 + 0.2
    x2 = prng.normal(41, "ap.x2", (B, c2, H, W)) * 0.7 - 0.1
    g = prng.normal(41, "ap.g", (c1 + c2,), 0.1, 1.0); b = prng.normal(41, "ap.b", (c1 + c2,), 0.1)
    ref = F.silu(F.group_norm(torch.cat([x1, x2], 1), 32, g, b, 1e-5))
    d1, d2 = nhwc(x1).to(dev), nhwc(x2).to(dev)
    cs1 = torch.empty((B, 5, c1, 2), device=dev); cs2 = torch.empty((B, 8, c2, 2), device=dev)     # 5 runs of 52 px; 8 runs of 32 px
    ops.gn_chan_stats(d1, cs1); ops.gn_chan_stats(d2, cs2)
    assert torch.allclose(cs1.sum(1)[..., 0].cpu(), x1.sum((2, 3)), rtol=1e-5, atol=1e-3)
    hi = torch.empty((B, H, W, c1 + c2), dtype=torch.int16, device=dev); lo = torch.empty_like(hi)
    ops.gn_apply16c(d1, cs1, d2, cs2, hi, lo, prec, g.to(dev), b.to(dev), 1e-5, 32, 1)
    assert rel_err(nchw(_as_float(hi, prec) + _as_float(lo, prec)), ref) < 2e-5


@pytest.mark.parametrize("B,H,W,c1,c2,cout", [(4, 32, 32, 4, 3, 128), (2, 16, 16, 7, 0, 32), (3, 8, 8, 4, 3, 64)])
def test_conv_in_epilogue_chan_stats(dev, B, H, W, c1, c2, cout):
    from stedm_amd import ops
    x1 = torch.randn(B, c1, H, W, device=dev); x2 = torch.randn(B, c2, H, W, device=dev) if c2 else None
    w = torch.randn(cout, c1 + c2, 3, 3, device=dev) * 0.2; bias = torch.randn(cout, device=dev)
    out = torch.empty(B, H, W, cout, device=dev)
    cs = torch.full((B, H // 2, cout, 2), float("nan"), device=dev)
    assert ops.conv_in(x1, x2, w, bias, out, chan_stats=cs)
    ref = F.conv2d(x1 if x2 is None else torch.cat([x1, x2], 1), w, bias, padding=1)
    assert rel_err(nchw(out), ref.cpu()) < 2e-5
    rows = out.view(B, H // 2, 2 * W, cout).double()
    assert torch.allclose(cs[..., 0].double(), rows.sum(2), rtol=1e-5, atol=1e-3)
    assert torch.allclose(cs[..., 1].double(), (rows * rows).sum(2), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("prec", ["f16", "bf16"])
@pytest.mark.parametrize("B,H,W,c,cout", [(128, 8, 8, 64, 128), (50, 16, 16, 128, 96), (13, 32, 32, 64, 160)])
def test_conv_up_subpixel_epilogue_chan_stats(dev, prec, B, H, W, c, cout):
    """statistics of the upsampled tensor in 4 * ceil(H*W/256) slots per sample (low-res run x output parity)."""
    from stedm_amd import ops
    from stedm_amd._lib import CONV_UP_SUBPIXEL
    pr = ops.Precision.parse(prec)
    x = torch.randn(B, H, W, c, device=dev); w = torch.randn(cout, c, 3, 3, device=dev) / math.sqrt(c * 9)
    h16 = torch.empty((B, H, W, c), dtype=torch.int16, device=dev)
    ops.gn_apply16(x, None, h16, None, pr)
    whi, wlo = ops.pack_conv_weight_up(w, pr)
    out = torch.empty(B, 2 * H, 2 * W, cout, device=dev)
    cs = torch.full((B, 4 * ops.gn_chan_nslab(H * W), cout, 2), float("nan"), device=dev)
    ops.conv_igemm(None, whi, wlo, out, prec=pr, mode=CONV_UP_SUBPIXEL, src16=(h16, None), w_frag=ops.pack_conv_weight_up_frag(w, pr), chan_stats=cs)
    flat = out.view(B, -1, cout).double()
    assert torch.allclose(cs.sum(1)[..., 0].double(), flat.sum(1), rtol=1e-4, atol=5e-3)
    assert torch.allclose(cs.sum(1)[..., 1].double(), (flat * flat).sum(1), rtol=1e-4, atol=5e-3)


def test_pack_conv_weight_frag_layout(dev):
    from stedm_amd import ops
    prec = ops.Precision.parse("f16")
    cout, cin = 160, 48
    w = prng.normal(3, "wf.w", (cout, cin, 3, 3))
    got = ops.pack_conv_weight_frag(w.to(dev), prec).cpu().view(torch.float16).float()      # [tn][chunk][tap][q][lane][8]
    tn, ch, tap, q, lane, e = torch.meshgrid(*[torch.arange(n) for n in got.shape], indexing="ij")
    n = tn * 128 + (q // 2) * 64 + (q % 2) * 32 + (lane % 32)
    ci = ch * 16 + (lane // 32) * 8 + e
    wp = torch.cat([w, torch.zeros(256 - cout, cin, 3, 3)]).reshape(256, cin, 9)
    want = wp[n, ci, tap].half().float()
    assert torch.equal(got, want)


@pytest.mark.parametrize("prec,tol", PRECS[:2])
@pytest.mark.parametrize("B,H,W,c,mode", [(2, 8, 8, 64, "up"), (16, 16, 16, 128, "up"), (3, 4, 4, 32, "up"), (1, 32, 32, 32, "up"),
                                          (2, 16, 16, 64, "down"), (8, 32, 32, 128, "down"), (3, 8, 8, 32, "down")])
def test_conv_dma_updown(dev, prec, tol, B, H, W, c, mode):
    _conv_dma_case(dev, prec, tol, B, H, W, c, c, mode, 3, use_emb=False, use_res=False)


@pytest.mark.parametrize("prec,tol", PRECS[:2])
@pytest.mark.parametrize("B,H,W,c,cout", [(2, 8, 8, 64, 64), (16, 16, 16, 128, 96), (3, 4, 4, 32, 32), (1, 32, 32, 32, 64), (5, 8, 8, 128, 128)])
def test_conv_dma_up_subpixel(dev, prec, tol, B, H, W, c, cout):
    """nearest x2 + 3x3 evaluated as four parity 2x2 convs with pre-summed taps (STEDM_CONV_UP_SUBPIXEL)."""
    _conv_dma_case(dev, prec, tol, B, H, W, c, cout, "up2", 3, use_emb=True, use_res=True)


@pytest.mark.parametrize("prec,tol", PRECS)
@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 8, 8, 192, 128), (3, 4, 4, 128, 384), (1, 10, 10, 64, 64), (32, 8, 8, 256, 128)])
def test_conv_dma_1x1(dev, prec, tol, B, H, W, cin, cout):
    _conv_dma_case(dev, prec, tol, B, H, W, cin, cout, "s1", 1, use_emb=False)


def test_pack_frag_multi_equals_single_packs(dev):
    """ops.PackPlan: the recorded fragment-order packs (forward order, flipped / transposed dgrad order, both MFMA fragment orders, 3x3 and
    1x1, a cout that is not a multiple of 128) re-run as ONE launch give bit-identical outputs, also after the weights changed."""
    from stedm_amd import ops
    from stedm_amd.ops import Precision
    for mode in ("bf16", "f16"):
        prec = Precision.parse(mode)
        plan = ops.PackPlan(prec)
        ws, outs, refs = [], [], []
        for i, (co, ci, ks) in enumerate([(128, 128, 3), (192, 64, 3), (512, 256, 3), (64, 128, 1), (320, 64, 1)]):
            w = torch.randn(co, ci, ks, ks, device=dev)
            ws.append(w)
            taps = ks * ks
            outs.append(plan.frag_oihw(w, False))                                              # forward, 32x32x16 order
            if ci % 32 == 0:
                outs.append(plan.frag_oihw(w, True))                                           # forward, 16x16x32 order
            if co % 16 == 0:
                outs.append(plan.frag(w, taps, ci * taps, True, ci, co, ks, False))            # dgrad: flipped, transposed
            if ks == 3 and co % 32 == 0:
                outs.append(plan.frag(w, taps, ci * taps, True, ci, co, 3, True))

        def single():
            r = []
            for w in ws:
                co, ci, ks, _ = w.shape
                taps = ks * ks
                r.append(ops.pack_conv_weight_frag(w, prec))
                if ci % 32 == 0:
                    r.append(ops.pack_conv_weight_frag16(w, prec))
                if co % 16 == 0:
                    r.append(ops.pack_conv_weight_strided(w, taps, ci * taps, True, ci, co, ks, prec, want_hi=False, want_frag=True)[2])
                if ks == 3 and co % 32 == 0:
                    r.append(ops.pack_conv_weight_frag16(w, prec, sn=taps, sc=ci * taps, flip=True, cout=ci, cin=co, ks=3))
            return r

        for a, b in zip(outs, single()):
            assert torch.equal(a, b)
        for w in ws:                      # an "optimizer step": same storage, new values
            w.mul_(0.5).add_(0.1)
        for o in outs:
            o.zero_()
        plan.run()
        for a, b in zip(outs, single()):
            assert a.shape == b.shape and torch.equal(a, b)


@pytest.mark.parametrize("P,cin,cout", [(64 * 64, 128, 128), (64 * 256, 1536, 512), (4096, 1024, 3072), (64 * 1024, 256, 128)])
def test_wgrad1x1_direct_vs_einsum(dev, P, cin, cout):
    """direct weight gradient of a 1x1 convolution from the flat bf16 planes (split-K partials folded by wgrad_to_oihw) against the fp32
    contraction over the same bf16-rounded operands."""
    from stedm_amd import ops
    from stedm_amd.ops import Precision
    prec = Precision.parse("bf16")
    x = torch.randn(P, cin, device=dev).to(torch.bfloat16)
    dy = (torch.randn(P, cout, device=dev) * 0.1).to(torch.bfloat16)
    ns = ops.wgrad1x1_plan(P, cin, cout)
    assert ns > 0
    part = torch.empty(ns * cin * cout, device=dev)
    ops.wgrad1x1(x.view(torch.int16), dy.view(torch.int16), part, prec)
    grad = torch.empty(cout, cin, 1, 1, device=dev)
    ops.wgrad_to_oihw(part, grad, cin, cout, False, ns)
    want = dy.float().T @ x.float()
    err = float((grad.view(cout, cin) - want).abs().max() / want.std())
    print(f"[wgrad1x1 P={P} {cin}->{cout}, {ns} slices] max|diff|/std {err:.2e}")
    assert err < 1e-4
    assert ops.wgrad1x1_plan(P, 96, cout) == 0 and ops.wgrad1x1_plan(P + 1, cin, cout) == 0
