"""Style path with the reference's module surface, HIP-backed.

  sViT, SPT, LSA, Transformer, PreNorm, FeedForward   networks/vit_set.py (state-dict names kept)
  Agg_Linear / Agg_Max / Agg_Mean / Agg_None          networks/agg_blocks.py
  SpatialRescaler                                      ldm/modules/encoders/modules.py:104-133

torch.nn modules are parameter containers; `forward` launches HIP kernels through the C ABI.
The embedder of Agg_* is any nn.Module mapping [(b n), 3, h, w] -> [(b n), f]; the reference's swin_v2_t is built on the HIP kernels in
stedm_amd/swin.py (third-party torchvision arithmetic, SURVEY.md §8c: parity unpinned).
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from ._lib import StedmHipError
from .ops import Precision


def pair(t):
    return t if isinstance(t, tuple) else (t, t)


class PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim, dropout=0.):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden_dim), nn.GELU(), nn.Dropout(dropout), nn.Linear(hidden_dim, dim), nn.Dropout(dropout))


class LSA(nn.Module):
    """vit_set.py:35-67 (container)."""

    def __init__(self, dim, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        inner_dim = dim_head * heads
        self.heads = heads
        self.dim_head = dim_head
        self.temperature = nn.Parameter(torch.log(torch.tensor(dim_head ** -0.5)))
        self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, dim), nn.Dropout(dropout))


class Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.):
        super().__init__()
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([PreNorm(dim, LSA(dim, heads=heads, dim_head=dim_head, dropout=dropout)),
                                              PreNorm(dim, FeedForward(dim, mlp_dim, dropout=dropout))]))


class SPT(nn.Module):
    """vit_set.py:84-107 (container; index 0 stands for the parameter-free Rearrange)."""

    def __init__(self, *, dim, patch_size, channels=3, sample_size=5):
        super().__init__()
        patch_dim = patch_size * patch_size * sample_size * channels
        self.to_patch_tokens = nn.Sequential(nn.Identity(), nn.LayerNorm(patch_dim), nn.Linear(patch_dim, dim))


class sViT(nn.Module):
    """Set-ViT style encoder, vit_set.py:109-208. forward(img [B, ns, H, W, 3]) -> [B, num_classes]."""

    def __init__(self, *, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim, pool='cls', channels=3, dim_head=64,
                 dropout=0., emb_dropout=0., ns=5, t_dim=256, precision: str = "parity"):
        super().__init__()
        image_height, image_width = pair(image_size)
        patch_height, patch_width = pair(patch_size)
        assert image_height % patch_height == 0 and image_width % patch_width == 0, \
            'Image dimensions must be divisible by the patch size.'
        assert pool in {'cls', 'mean', 'none'}, 'pool type must be either cls (cls token) or mean (mean pooling)'
        if pool == 'none':
            raise NotImplementedError("pool='none' (per-token output) is not used by STEDM (svit.yaml: pool mean)")
        if dim_head != 64 or channels != 3 or patch_height != patch_width:
            raise NotImplementedError("HIP sViT: dim_head 64, 3 channels and square patches only (conf/style_agg/svit.yaml)")
        self.ns = ns
        self.np = (image_height // patch_height) * (image_width // patch_width)
        self.patch_size = patch_height
        self.dim, self.heads, self.depth = dim, heads, depth
        self.to_patch_embedding = SPT(dim=dim, patch_size=patch_height, channels=channels, sample_size=ns)
        self.pos_embedding = nn.Parameter(torch.randn(1, self.np + 2, dim))
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
        self.dropout = nn.Dropout(emb_dropout)
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, dropout)
        self.pool = pool
        self.to_latent = nn.Identity()
        self.mlp_head = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, num_classes))
        self.to_time_embedding = nn.Linear(t_dim, dim)
        self.precision = Precision.parse(precision) if isinstance(precision, str) else precision
        self._packed: Dict = {}
        self._pack_key = None
        self._bufs: Dict[Tuple, torch.Tensor] = {}
        # train-mode dropout: a forward draws its seed from torch's generator (so torch.manual_seed governs it) unless `dropout_seed` is set;
        # `last_dropout_seed` is what the last forward used (tests rebuild the masks from it)
        self.dropout_seed: Optional[int] = None
        self.last_dropout_seed = 0
        # to_qkv's epilogue writes the attention's operand planes (single-product modes, no dropout); STEDM_SVIT_FUSE_QKV=0: the separate pack pass
        self.fuse_qkv = os.environ.get("STEDM_SVIT_FUSE_QKV", "1") != "0"

    # mask stream ids of the dropout sites (include/stedm_hip.h, "train-mode dropout"): site = 8 * layer + kind, the embedding site alone
    SITE_EMB, SITE_ATTN, SITE_OUT, SITE_FF1, SITE_FF2 = 0x10000, 1, 2, 3, 4

    # ---------------------------------------------------------------------------------------------- engine
    def set_precision(self, precision):
        self.precision = Precision.parse(precision) if isinstance(precision, str) else precision
        self._pack_key = None

    def _buf(self, name, shape, dtype=torch.float32, zero=False):
        key = (name, tuple(shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            dev = self.pos_embedding.device
            t = (torch.zeros if zero else torch.empty)(tuple(shape), dtype=dtype, device=dev)
            self._bufs[key] = t
        return t

    def _prepare(self):
        params = list(self.parameters())
        dev = params[0].device
        if dev.type != "cuda":
            raise StedmHipError("sViT.forward needs its parameters on the GPU; there is no CPU fallback")
        key = (self.precision, dev, tuple(p._version for p in params), tuple(p.data_ptr() for p in params))
        if key == self._pack_key:
            return
        P = {}
        prec = self.precision
        tp = self.to_patch_embedding.to_patch_tokens
        def pack(w):   # Linear [N][K] -> packed 1x1 conv operands (+ fragment order for the register-streamed kernel)
            w4 = w.float().unsqueeze(-1).unsqueeze(-1)
            hi, lo = ops.pack_conv_weight(w4, prec)
            frag = ops.pack_conv_weight_frag(w4, prec) if prec.npass == 1 and w4.shape[1] % 64 == 0 else None
            return hi, lo, frag

        P["pe"] = pack(tp[2].weight)
        for l, (attn, ff) in enumerate(self.transformer.layers):
            P[f"qkv{l}"] = pack(attn.fn.to_qkv.weight)
            P[f"out{l}"] = pack(attn.fn.to_out[0].weight)
            P[f"ff1{l}"] = pack(ff.fn.net[0].weight)
            P[f"ff2{l}"] = pack(ff.fn.net[3].weight)
            # logits scale exp(temperature) (vit_set.py:56) times log2(e): lsa_flash exponentiates with the hardware exp2
            P[f"tau{l}"] = float(attn.fn.temperature.detach().exp().item()) * 1.4426950408889634
        P["head_wt"] = ops.transpose(self.mlp_head[1].weight.float())
        self._packed = P
        self._pack_key = key

    def _gemm(self, a16, w, M, N, bias=None, res=None, out=None, act_out=0, out16=None, qkv_planes=None, query_rs=False):
        """[M, K] 16-bit planes x packed [N][1][K] weights on the DMA conv kernel (1x1 conv view [1, 1, M, K])."""
        K = a16[0].shape[-1]
        v = lambda t: None if t is None else t.view(1, 1, M, -1)
        return ops.conv_igemm(None, w[0], w[1], v(out), prec=self.precision, ks=1, src16=(v(a16[0]), v(a16[1])), bias=bias, res=v(res),
                              act_out=act_out, out16=None if out16 is None else (v(out16[0]), v(out16[1])), w_frag=w[2],
                              qkv_planes=qkv_planes, query_rs=query_rs)

    @torch.no_grad()
    def forward(self, img, t_emb=None, c_old=None):
        """vit_set.py:165-208. img [B, ns, H, W, 3] fp32 (NHWC per image, as LDM_Diffusion.prepare_batch emits it)."""
        if t_emb is not None:
            raise NotImplementedError("t_emb is always None in STEDM (networks/s_zss_dm.py:55)")
        # train mode (the reference runs the agg block inside the training step: networks/s_zss_dm.py:45-60): nn.Dropout is live after the
        # pos_embedding (vit_set.py:187), on the attention probabilities (:62), after to_out (:49) and twice in the FeedForward (:28-30)
        p_emb = float(self.dropout.p) if self.training else 0.0
        p_drop = float(self.transformer.layers[0][0].fn.to_out[1].p) if self.training and self.depth else 0.0
        seed = 0
        if p_emb > 0 or p_drop > 0:
            if self.precision.attn_fp8:
                raise NotImplementedError("train-mode dropout is not built for the fp8 attention experiment (prediction-only mode)")
            seed = self.dropout_seed if self.dropout_seed is not None else int(torch.randint(0, 2 ** 62, (1,)).item())
        self.last_dropout_seed = seed
        self._prepare()
        P, prec = self._packed, self.precision
        img = img.float().contiguous()
        B, ns, H, W, _ = img.shape
        assert ns == self.ns, f"sViT built for ns={self.ns} style images, got {ns}"
        dim, heads = self.dim, self.heads
        n = (H // self.patch_size) * (W // self.patch_size)
        assert n == self.np, "image size does not match the sViT's pos_embedding"
        T = n + 2
        Tp = ((T + 127) // 128) * 128
        M = B * T
        tp = self.to_patch_embedding.to_patch_tokens
        x = self._buf("x", (B, T, dim))
        i16 = torch.int16
        lo_ok = prec.npass == 3
        # SPT: gather + LayerNorm -> 16-bit planes, Linear on MFMA, + pos_embedding / cls / zero time token
        pd = tp[2].in_features
        pe16 = (self._buf("pe.hi", (B * n, pd), i16), self._buf("pe.lo", (B * n, pd), i16) if lo_ok else None)
        ops.svit_patch_ln16(img, tp[1].weight, tp[1].bias, tp[1].eps, pe16[0], pe16[1], self.patch_size, prec)
        tok = self._buf("pe.tok", (B * n, dim))
        self._gemm(pe16, P["pe"], B * n, dim, bias=tp[2].bias, out=tok)
        ops.svit_tok_place(tok, self.pos_embedding, self.cls_token, x)
        if p_emb > 0:
            ops.dropout_rows(x, p_emb, seed, self.SITE_EMB, prec, out=x)
        ln = (self._buf("ln.hi", (M, dim), i16), self._buf("ln.lo", (M, dim), i16) if lo_ok else None)
        # single-product modes: to_qkv writes its 16-bit output only (the attention's operands are 16-bit — or MX-fp8 packed from them — anyway)
        q16 = prec.npass == 1
        mk = lambda nm, shp: (self._buf(nm + ".hi", shp, i16, zero=True), self._buf(nm + ".lo", shp, i16, zero=True) if lo_ok else None)
        q, k, vt = mk("q", (B * heads, Tp, 64)), mk("k", (B * heads, Tp, 64)), mk("vt", (B * heads, 64, Tp))
        att = (self._buf("att.hi", (M, heads * 64), i16), self._buf("att.lo", (M, heads * 64), i16) if lo_ok else None)
        # ... and, 16-bit operands without dropout: the GEMM's epilogue writes the attention's q / k / v^T planes itself (stedm_conv_args.qkv_*:
        # no [M][3 * heads * 64] tensor, no qkv_pack pass), where the register-streamed kernel runs the problem
        planes = lambda l: (q[0], k[0], vt[0], T, Tp, heads, P[f"tau{l}"])
        fused_qkv = (q16 and not prec.attn_fp8 and p_drop == 0 and self.fuse_qkv and self.depth > 0 and T % 2 == 0 and heads % 2 == 0
                     and bool(self._gemm(ln, P["qkv0"], M, 3 * heads * 64, qkv_planes=planes(0), query_rs=True)))
        qkv = None if fused_qkv else (self._buf("qkv16", (M, 3 * heads * 64), i16) if q16 else self._buf("qkv", (M, 3 * heads * 64)))
        for l, (attn, ff) in enumerate(self.transformer.layers):
            mlp = ff.fn.net[0].out_features
            ops.ln_apply16(x, attn.norm.weight, attn.norm.bias, attn.norm.eps, ln[0], ln[1], prec)
            if fused_qkv:
                self._gemm(ln, P[f"qkv{l}"], M, 3 * heads * 64, qkv_planes=planes(l))
            elif q16:
                self._gemm(ln, P[f"qkv{l}"], M, 3 * heads * 64, out16=(qkv, None))
            else:
                self._gemm(ln, P[f"qkv{l}"], M, 3 * heads * 64, out=qkv)
            if prec.attn_fp8:     # MX-fp8 operands: e4m3 bytes + one E8M0 scale per 32 elements (BASELINE config 5)
                u8 = torch.uint8
                q8, k8, v8 = (self._buf(nm, shp, u8, zero=True) for nm, shp in (("q8", (B * heads, Tp, 64)), ("k8", (B * heads, Tp, 64)), ("vt8", (B * heads, 64, Tp))))
                qs, ks, vs = (self._buf(nm, shp, u8, zero=True) for nm, shp in (("qs8", (B * heads, Tp, 2)), ("ks8", (B * heads, Tp, 2)), ("vs8", (B * heads, Tp // 32, 64))))
                ops.qkv_pack_mx8(qkv, P[f"tau{l}"], q8, qs, k8, ks, v8, vs, B, T, Tp, heads, prec)
                ops.lsa_flash_mx8(q8, qs, k8, ks, v8, vs, att[0], B, T, Tp, heads, prec)
            elif p_drop > 0:
                ops.qkv_pack(qkv, P[f"tau{l}"], q, k, vt, B, T, Tp, heads, prec)
                ops.lsa_flash_drop(q, k, vt, att, B, T, Tp, heads, prec, p_drop, seed, 8 * l + self.SITE_ATTN)
            else:
                if not fused_qkv:
                    ops.qkv_pack(qkv, P[f"tau{l}"], q, k, vt, B, T, Tp, heads, prec)
                ops.lsa_flash(q, k, vt, att, B, T, Tp, heads, prec)
            h16 = (self._buf("h.hi", (M, mlp), i16), self._buf("h.lo", (M, mlp), i16) if lo_ok else None)
            if p_drop > 0:
                # the Linear's output goes to a scratch tensor, the dropout kernel masks it and adds the residual / writes the operand planes
                y = self._buf("drop.y", (M, max(dim, mlp)))
                yd, ym = y.view(-1)[:M * dim].view(M, dim), y.view(-1)[:M * mlp].view(M, mlp)
                self._gemm(att, P[f"out{l}"], M, dim, bias=attn.fn.to_out[0].bias, out=yd)
                ops.dropout_rows(yd, p_drop, seed, 8 * l + self.SITE_OUT, prec, res=x, out=x)                  # x = drop(attn(x)) + x
                ops.ln_apply16(x, ff.norm.weight, ff.norm.bias, ff.norm.eps, ln[0], ln[1], prec)
                self._gemm(ln, P[f"ff1{l}"], M, mlp, bias=ff.fn.net[0].bias, act_out=2, out=ym)                # GELU(Linear)
                ops.dropout_rows(ym, p_drop, seed, 8 * l + self.SITE_FF1, prec, hi=h16[0], lo=h16[1])
                self._gemm(h16, P[f"ff2{l}"], M, dim, bias=ff.fn.net[3].bias, out=yd)
                ops.dropout_rows(yd, p_drop, seed, 8 * l + self.SITE_FF2, prec, res=x, out=x)                  # x = drop(ff(x)) + x
                continue
            self._gemm(att, P[f"out{l}"], M, dim, bias=attn.fn.to_out[0].bias, res=x, out=x)          # x = attn(x) + x
            ops.ln_apply16(x, ff.norm.weight, ff.norm.bias, ff.norm.eps, ln[0], ln[1], prec)
            self._gemm(ln, P[f"ff1{l}"], M, mlp, bias=ff.fn.net[0].bias, act_out=2, out16=h16)       # GELU(Linear)
            self._gemm(h16, P[f"ff2{l}"], M, dim, bias=ff.fn.net[3].bias, res=x, out=x)              # x = ff(x) + x
        out = torch.empty((B, self.mlp_head[1].out_features), dtype=torch.float32, device=img.device)
        pool = {"mean": 0, "cls": 1, "sum": 2}[self.pool]
        ops.svit_head(x, pool, None if c_old is None else c_old.float().contiguous(), self.mlp_head[0].weight, self.mlp_head[0].bias,
                      self.mlp_head[0].eps, P["head_wt"], self.mlp_head[1].bias, out, ws=self._buf("head.ws", (1024 * dim,)))
        return out


# ---------------------------------------------------------------------------------------------------- agg blocks
class _AggBase(nn.Module):
    def set_precision(self, precision):
        """MFMA operand mode of a HIP-backed embedder (stedm_amd.swin); a foreign embedder keeps its own arithmetic."""
        emb = getattr(self, "_embedder", None)
        if emb is not None and hasattr(emb, "set_precision"):
            emb.set_precision(precision)


def _embed(embedder: nn.Module, style_imgs: torch.Tensor) -> torch.Tensor:
    """'b n h w c -> (b n) c h w' then the caller's embedder -> [(b n), f] (agg_blocks.py:26-28)."""
    b, n, h, w, c = style_imgs.shape
    x = style_imgs.permute(0, 1, 4, 2, 3).reshape(b * n, c, h, w)
    return embedder(x).float().contiguous()


class Agg_Linear(_AggBase):
    """agg_blocks.py:6-33."""

    def __init__(self, sampling_cfg, embedder):
        super().__init__()
        self._sampling_cfg = sampling_cfg
        self._embedder = embedder
        num = self._sampling_cfg.num_patches if self._sampling_cfg.name == "mp" else 1
        self._linear_block = nn.Sequential(nn.ReLU(), nn.Linear(512 * num, 512), nn.ReLU(), nn.Linear(512, 512), nn.ReLU())
        self.register_module("embedder", self._embedder)
        self.register_module("linear_block", self._linear_block)

    @torch.no_grad()
    def forward(self, style_imgs):
        b = style_imgs.shape[0]
        f = _embed(self._embedder, style_imgs).reshape(b, -1)        # '(b1 n) f -> b1 (n f)'
        l1, l2 = self._linear_block[1], self._linear_block[3]
        h = ops.linear(f, ops.transpose(l1.weight.float()), l1.bias, torch.empty((b, 512), device=f.device), act_in=2, act_out=2)
        return ops.linear(h, ops.transpose(l2.weight.float()), l2.bias, torch.empty((b, 512), device=f.device), act_in=0, act_out=2)


class Agg_Max(_AggBase):
    """agg_blocks.py:36-54."""

    def __init__(self, sampling_cfg, embedder):
        super().__init__()
        self._sampling_cfg = sampling_cfg
        self._embedder = embedder
        self.register_module("embedder", self._embedder)

    @torch.no_grad()
    def forward(self, style_imgs):
        b, n = style_imgs.shape[:2]
        f = _embed(self._embedder, style_imgs)
        return ops.agg_reduce(f, torch.empty((b, f.shape[-1]), device=f.device), n, 1)


class Agg_Mean(_AggBase):
    """agg_blocks.py:57-75."""

    def __init__(self, sampling_cfg, embedder):
        super().__init__()
        self._sampling_cfg = sampling_cfg
        self._embedder = embedder
        self.register_module("embedder", self._embedder)

    @torch.no_grad()
    def forward(self, style_imgs):
        b, n = style_imgs.shape[:2]
        f = _embed(self._embedder, style_imgs)
        return ops.agg_reduce(f, torch.empty((b, f.shape[-1]), device=f.device), n, 0)


class Agg_None(_AggBase):
    """agg_blocks.py:78-86."""

    def __init__(self, sampling_cfg, embedder):
        super().__init__()
        self._sampling_cfg = sampling_cfg
        self._embedder = embedder

    def forward(self, style_imgs):
        return torch.zeros((style_imgs.shape[0], 512), dtype=style_imgs.dtype, device=style_imgs.device)


# ---------------------------------------------------------------------------------------------------- layout conditioner
class SpatialRescaler(nn.Module):
    """encoders/modules.py:104-133 — n_stages x bilinear 1/2, then a bias-free 1x1 conv (channel_mapper)."""

    def __init__(self, n_stages=1, method='bilinear', multiplier=0.5, in_channels=3, out_channels=None, bias=False):
        super().__init__()
        self.n_stages = n_stages
        assert self.n_stages >= 0
        assert method in ['nearest', 'linear', 'bilinear', 'trilinear', 'bicubic', 'area']
        if method != 'bilinear' or multiplier != 0.5 or bias:
            raise NotImplementedError("HIP SpatialRescaler: bilinear x0.5 stages without bias (conf/diffusion/cond_stage_config/spatial.yaml)")
        self.multiplier = multiplier
        self.in_channels = in_channels
        self.remap_output = out_channels is not None
        if self.remap_output:
            self.channel_mapper = nn.Conv2d(in_channels, out_channels, 1, bias=bias)

    @torch.no_grad()
    def forward(self, x):
        x = x.float().contiguous()
        B, cin, H, W = x.shape
        cout = self.channel_mapper.out_channels if self.remap_output else cin
        f = 1 << self.n_stages
        out = torch.empty((B, cout, H // f, W // f), dtype=torch.float32, device=x.device)
        w = self.channel_mapper.weight.detach().float().reshape(cout, cin).contiguous() if self.remap_output else None
        return ops.spatial_rescale(x, w, out, self.n_stages)

    def encode(self, x):
        return self(x)

    @torch.no_grad()
    def backward(self, x, d_out, accumulate: bool = False):
        """Gradient of the only trainable tensor of the shipped cond stage (channel_mapper.weight, `cond_stage_trainable`) given
        d_out = dL/d(forward(x)) — the c_concat slice of the U-Net's input gradient (UNetTrainer.backward). Fills `.grad`."""
        if not self.remap_output:
            return None
        wt = self.channel_mapper.weight
        if wt.grad is None:
            wt.grad = torch.zeros_like(wt, dtype=torch.float32)
        ops.spatial_rescale_wgrad(x.float().contiguous(), d_out.float().contiguous(), wt.grad, self.n_stages, accumulate)
        return wt.grad
