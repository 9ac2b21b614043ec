"""Swin-Transformer-V2 style embedder with torchvision's module surface, HIP-backed.

The reference builds `torchvision.models.get_model("swin_v2_t")` and swaps its head for `Linear(768, 512)` (networks/s_zss_dm.py:19-20);
the Agg_* blocks call it on '(b n) c h w' images (networks/agg_blocks.py:28,49,70). torchvision (pinned 0.18.1) is third-party and not
under /root/reference: this file restates the published architecture (parity unpinned: DESIGN.md §2) with torchvision's state-dict
names, so a torchvision checkpoint of swin_v2_t / swin_v2_s / swin_v2_b loads with `load_state_dict`:

  features.0.{0 Conv2d(3, C, 4, 4) | 2 LayerNorm}         features.{1,3,5,7}.<i>.{norm1, attn.{qkv, proj, logit_scale, cpb_mlp.0, cpb_mlp.2,
  features.{2,4,6}.{reduction, norm} (PatchMergingV2)       relative_coords_table, relative_position_index}, norm2, mlp.0, mlp.3}
  norm, head

torch.nn modules are parameter containers; `forward` launches HIP kernels through the C ABI: every Linear (and the patch conv) is a
`stedm_conv_igemm` 1x1 GEMM over 16-bit operand planes, the rest is csrc/swin.hip. No CPU fallback.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from ._lib import StedmHipError
from .ops import Precision

WINDOW = 8          # swin_v2_{t,s,b}: window_size [8, 8]
HEAD_DIM = 32       # embed_dim / num_heads of all three


class ShiftedWindowAttentionV2(nn.Module):
    """Container of torchvision's ShiftedWindowAttentionV2 (cosine attention, log-spaced continuous position bias)."""

    def __init__(self, dim: int, window_size: List[int], shift_size: List[int], num_heads: int):
        super().__init__()
        self.window_size, self.shift_size, self.num_heads = list(window_size), list(shift_size), num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim, bias=True)
        self.logit_scale = nn.Parameter(torch.log(10 * torch.ones((num_heads, 1, 1))))
        self.cpb_mlp = nn.Sequential(nn.Linear(2, 512, bias=True), nn.ReLU(inplace=True), nn.Linear(512, num_heads, bias=False))
        with torch.no_grad():   # the key bias is identically zero in V2 (zeroed at construction and again in every forward)
            n = self.qkv.bias.numel() // 3
            self.qkv.bias[n:2 * n].zero_()
        # log-spaced relative coordinates, normalised to [-8, 8] before the log: sign(x) log2(|x| + 1) / log2(8)
        ws = self.window_size
        ch = torch.arange(-(ws[0] - 1), ws[0], dtype=torch.float32)
        cw = torch.arange(-(ws[1] - 1), ws[1], dtype=torch.float32)
        table = torch.stack(torch.meshgrid([ch, cw], indexing="ij")).permute(1, 2, 0).contiguous().unsqueeze(0)   # [1, 2Wh-1, 2Ww-1, 2]
        table[:, :, :, 0] /= ws[0] - 1
        table[:, :, :, 1] /= ws[1] - 1
        table *= 8
        table = torch.sign(table) * torch.log2(torch.abs(table) + 1.0) / 3.0
        self.register_buffer("relative_coords_table", table)
        # pair-wise relative position index of the tokens inside a window
        coords = torch.stack(torch.meshgrid(torch.arange(ws[0]), torch.arange(ws[1]), indexing="ij")).flatten(1)   # [2, Wh*Ww]
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += ws[0] - 1
        rel[:, :, 1] += ws[1] - 1
        rel[:, :, 0] *= 2 * ws[1] - 1
        self.register_buffer("relative_position_index", rel.sum(-1).flatten())


class SwinTransformerBlockV2(nn.Module):
    """x = x + sd(norm1(attn(x))); x = x + sd(norm2(mlp(x))) (post-norm; sd = stochastic depth: per-image gates in train mode — drawn by
    SwinTransformerV2.forward — and the identity in eval mode)."""

    def __init__(self, dim: int, num_heads: int, window_size: List[int], shift_size: List[int], mlp_ratio: float = 4.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn = ShiftedWindowAttentionV2(dim, window_size, shift_size, num_heads)
        self.stochastic_depth = nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        hidden = int(dim * mlp_ratio)
        self.mlp = nn.Sequential(nn.Linear(dim, hidden), nn.GELU(), nn.Dropout(0.0), nn.Linear(hidden, dim), nn.Dropout(0.0))


class PatchMergingV2(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(2 * dim, eps=1e-5)


class SwinTransformerV2(nn.Module):
    """torchvision.models.swin_transformer.SwinTransformer with the V2 block / merging layers. forward(x [N, 3, H, W]) -> [N, classes]."""

    def __init__(self, embed_dim: int = 96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window_size=(WINDOW, WINDOW),
                 mlp_ratio: float = 4.0, num_classes: int = 1000, precision: str = "parity", chunk_images: int = 32,
                 stochastic_depth_prob: float = 0.0):
        super().__init__()
        if tuple(window_size) != (WINDOW, WINDOW) or any(embed_dim * 2 ** i != HEAD_DIM * h for i, h in enumerate(num_heads)):
            raise NotImplementedError("HIP Swin-V2: 8 x 8 windows and head dim 32 (swin_v2_t / swin_v2_s / swin_v2_b)")
        if embed_dim % 32:
            raise NotImplementedError("HIP Swin-V2: embed_dim must be a multiple of 32")
        self.embed_dim, self.depths, self.heads = embed_dim, tuple(depths), tuple(num_heads)
        layers: List[nn.Module] = [nn.Sequential(nn.Conv2d(3, embed_dim, kernel_size=4, stride=4), nn.Identity(), nn.LayerNorm(embed_dim, eps=1e-5))]
        for s, depth in enumerate(depths):
            dim = embed_dim * 2 ** s
            layers.append(nn.Sequential(*[SwinTransformerBlockV2(dim, num_heads[s], list(window_size),
                                                                  [0 if i % 2 == 0 else w // 2 for w in window_size], mlp_ratio) for i in range(depth)]))
            if s < len(depths) - 1:
                layers.append(PatchMergingV2(dim))
        self.features = nn.Sequential(*layers)
        nf = embed_dim * 2 ** (len(depths) - 1)
        self.norm = nn.LayerNorm(nf, eps=1e-5)
        self.permute = nn.Identity()
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.flatten = nn.Flatten(1)
        self.head = nn.Linear(nf, num_classes)
        for m in self.modules():   # torchvision's initialisation
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        self.precision = Precision.parse(precision) if isinstance(precision, str) else precision
        self.chunk_images = chunk_images
        # rows of <= 128 channels: the post-norm LayerNorm (+ residual) runs in the producing GEMM's epilogue; STEDM_SWIN_FUSE_LN=0: its own pass
        self.fuse_ln = os.environ.get("STEDM_SWIN_FUSE_LN", "1") != "0"
        # train-mode stochastic depth (torchvision: StochasticDepth(p_i, "row") on both residual branches of block i, p_i rising linearly to
        # stochastic_depth_prob over all blocks). `sd_gates` ([blocks, 2, N] fp32, = bernoulli(1 - p_i) / (1 - p_i)) overrides the draw (tests).
        nb = sum(depths)
        self.sd_probs = [stochastic_depth_prob * i / max(nb - 1.0, 1.0) for i in range(nb)]
        self.sd_gates: Optional[torch.Tensor] = None
        self._gates: Optional[torch.Tensor] = None
        self._packed: Dict = {}
        self._pack_key = None
        self._bufs: Dict[Tuple, torch.Tensor] = {}

    # ---------------------------------------------------------------------------------------------- engine
    def set_precision(self, precision):
        self.precision = Precision.parse(precision) if isinstance(precision, str) else precision
        self._pack_key = None

    def _buf(self, name, shape, dtype=torch.float32):
        key = (name, tuple(shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            t = torch.empty(tuple(shape), dtype=dtype, device=self.norm.weight.device)
            self._bufs[key] = t
        return t

    def _planes(self, name, shape):
        """16-bit operand planes [rows, K]; K is padded to the GEMM kernels' 64-channel chunks (zero columns, written once here: the producers
        only touch the first `dim` columns) — the register-streamed 1x1 kernel needs K % 64 == 0, and stage 1 of swin_v2_t has K = 96."""
        i16 = torch.int16
        rows, k = shape
        kp = (k + 63) // 64 * 64
        def mk(nm):
            key = (nm, (rows, kp), i16)
            t = self._bufs.get(key)
            if t is None:
                t = torch.zeros((rows, kp), dtype=i16, device=self.norm.weight.device)
                self._bufs[key] = t
            return t
        return (mk(name + ".hi"), mk(name + ".lo") if self.precision.npass == 3 else None)

    def _stages(self):
        """[(blocks, merge or None)] in order."""
        mods = list(self.features)[1:]
        out, i = [], 0
        while i < len(mods):
            merge = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], PatchMergingV2) else None
            out.append((list(mods[i]), merge))
            i += 2 if merge is not None else 1
        return out

    def _prepare(self):
        params = list(self.parameters())
        dev = params[0].device
        if dev.type != "cuda":
            raise StedmHipError("SwinTransformerV2.forward needs its parameters on the GPU; there is no CPU fallback")
        key = (self.precision, dev, tuple(p._version for p in params), tuple(p.data_ptr() for p in params))
        if key == self._pack_key:
            return
        prec = self.precision

        def pack(w):   # Linear [N][K] -> packed 1x1 conv operands (+ fragment order for the register-streamed kernel)
            w = w.detach().float()
            kpad = (w.shape[1] + 63) // 64 * 64          # zero columns up to the 64-channel chunk (the operand planes are padded alike)
            if kpad > w.shape[1]:
                w = torch.cat([w, w.new_zeros(w.shape[0], kpad - w.shape[1])], dim=1)
            w4 = w.contiguous().unsqueeze(-1).unsqueeze(-1)
            hi, lo = ops.pack_conv_weight(w4, prec)
            frag = ops.pack_conv_weight_frag(w4, prec) if prec.npass == 1 and w4.shape[1] % 64 == 0 else None
            return hi, lo, frag

        P: Dict = {}
        conv = self.features[0][0]
        P["pe"] = pack(conv.weight.reshape(conv.out_channels, -1))      # OIHW flattened: k = c*16 + ky*4 + kx
        for s, (blocks, merge) in enumerate(self._stages()):
            for i, blk in enumerate(blocks):
                at = blk.attn
                nm = f"{s}.{i}"
                P["qkv" + nm] = pack(at.qkv.weight)
                bz = at.qkv.bias.detach().float().clone()
                n = bz.numel() // 3
                bz[n:2 * n].zero_()                                               # shifted_window_attention zeroes the key bias
                P["qkvb" + nm] = bz
                P["proj" + nm] = pack(at.proj.weight)
                P["fc1" + nm] = pack(blk.mlp[0].weight)
                P["fc2" + nm] = pack(blk.mlp[3].weight)
                P["scale" + nm] = torch.clamp(at.logit_scale.detach().float().reshape(-1), max=math.log(100.0)).exp().contiguous()
                # continuous position bias: cpb_mlp over the (2Wh-1)(2Ww-1) table, gathered per token pair, 16 sigmoid
                table = at.relative_coords_table.detach().float().reshape(-1, 2).contiguous()
                l0, l2 = at.cpb_mlp[0], at.cpb_mlp[2]
                h1 = ops.linear(table, ops.transpose(l0.weight.detach().float().contiguous()), l0.bias.detach().float(),
                                torch.empty((table.shape[0], l0.out_features), device=dev), act_out=2)
                cpb = ops.linear(h1, ops.transpose(l2.weight.detach().float().contiguous()), None, torch.empty((table.shape[0], at.num_heads), device=dev))
                P["rpb" + nm] = ops.swin_rpb(cpb, at.relative_position_index.to(torch.int64).contiguous(), at.num_heads)
            if merge is not None:
                P[f"red{s}"] = pack(merge.reduction.weight)
        self._packed = P
        self._pack_key = key

    def _gemm(self, a16, w, M, bias=None, res=None, out=None, act_out=0, out16=None, ln_after=None, query_rs=False):
        """[M, K] 16-bit planes x packed [N][1][K] weights on the DMA conv kernels (1x1 conv view [1, 1, M, K])."""
        v = lambda t: None if t is None else t.view(1, 1, M, -1)
        return ops.conv_igemm(None, w[0], w[1], v(out), prec=self.precision, ks=1, src16=(v(a16[0]), v(a16[1])), bias=bias, res=v(res),
                              act_out=act_out, out16=None if out16 is None else (v(out16[0]), v(out16[1])), w_frag=w[2], ln_after=ln_after,
                              query_rs=query_rs)

    def _gemm_ln(self, a16, w, M, bias, norm, res, xc, x16, y, gate=None, rows_per_gate=1):
        """x = (res +) norm(a16 @ w^T + bias) -> fp32 xc and the operand planes x16. Rows of up to 128 channels (stage 1, the patch embedding)
        without a stochastic-depth gate: the GEMM's epilogue normalises the row itself (stedm_conv_args.ln_*: no fp32 GEMM output, no LayerNorm
        pass); otherwise the GEMM writes y and stedm_swin_ln follows."""
        prec = self.precision
        ln = (norm.weight, norm.bias, norm.eps, res)
        if (self.fuse_ln and gate is None and prec.npass == 1 and xc.shape[-1] <= 128 and xc.shape[-1] % 4 == 0
                and self._gemm(a16, w, M, bias=bias, out=xc, out16=(x16[0], None), ln_after=ln, query_rs=True)):
            self._gemm(a16, w, M, bias=bias, out=xc, out16=(x16[0], None), ln_after=ln)
            return
        self._gemm(a16, w, M, bias=bias, out=y)
        ops.swin_ln(y, norm.weight, norm.bias, norm.eps, res, xc, x16[0], x16[1], prec, gate=gate, rows_per_gate=rows_per_gate)

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"expected images [N, 3, H, W], got {tuple(x.shape)}")
        if x.dtype != torch.float32:
            x = x.float()
        self._prepare()
        N = x.shape[0]
        out = torch.empty((N, self.head.out_features), dtype=torch.float32, device=x.device)
        # train mode (the reference runs the embedder inside the training step in train mode, never optimising it: networks/s_zss_dm.py:45-60,
        # modules/ldm_diffusion.py:224-234): per-image gates of the residual branches; eval / p = 0: no gates, the eval arithmetic bit for bit
        self._gates = None
        if self.training and any(p > 0 for p in self.sd_probs):
            if self.sd_gates is not None:
                g = self.sd_gates.to(device=x.device, dtype=torch.float32)
                assert tuple(g.shape) == (len(self.sd_probs), 2, N), f"sd_gates must be [blocks, 2, N] = {(len(self.sd_probs), 2, N)}"
            else:
                surv = 1.0 - torch.tensor(self.sd_probs, dtype=torch.float32, device=x.device).view(-1, 1, 1)
                g = (torch.rand((len(self.sd_probs), 2, N), device=x.device) < surv).float() / surv
            self._gates = g.contiguous()
        for n0 in range(0, N, self.chunk_images):
            self._forward_chunk(x[n0:n0 + self.chunk_images], out[n0:n0 + self.chunk_images], n0)
        return out

    def _forward_chunk(self, x: torch.Tensor, out: torch.Tensor, n0: int = 0) -> None:
        P, prec = self._packed, self.precision
        N, _, Himg, Wimg = x.shape
        H, W = Himg // 4, Wimg // 4
        M = N * H * W
        dim = self.embed_dim
        pe16 = self._planes("pe", (M, 64))
        ops.swin_patch16(x, pe16[0], pe16[1], prec)
        y = self._buf("y0", (M, dim))
        conv, ln0 = self.features[0][0], self.features[0][2]
        xc = self._buf("x0", (M, dim))
        x16 = self._planes("x16.0", (M, dim))
        self._gemm_ln(pe16, P["pe"], M, conv.bias, ln0, None, xc, x16, y)
        bi = 0          # block index over all stages (stochastic-depth gate row)
        for s, (blocks, merge) in enumerate(self._stages()):
            heads = self.heads[s]
            # single-product modes: the attention's operands are rounded to 16 bits anyway, so the qkv GEMM writes its 16-bit output only
            # (stage 1's fp32 qkv tensor was the largest stream of the model: 18.9 MB per image and block)
            q16 = prec.npass == 1
            qkv = self._buf(f"qkv16.{s}", (M, 3 * dim), torch.int16) if q16 else self._buf(f"qkv{s}", (M, 3 * dim))
            att = self._planes(f"att{s}", (M, dim))
            y = self._buf(f"y{s}", (M, dim))
            for i, blk in enumerate(blocks):
                nm = f"{s}.{i}"
                at = blk.attn
                if q16:
                    self._gemm(x16, P["qkv" + nm], M, bias=P["qkvb" + nm], out16=(qkv, None))
                else:
                    self._gemm(x16, P["qkv" + nm], M, bias=P["qkvb" + nm], out=qkv)
                ops.swin_window_attn(qkv, P["qkvb" + nm], P["scale" + nm], P["rpb" + nm], att[0], att[1], N, H, W, heads, at.shift_size[0], prec)
                g1 = g2 = None
                if self._gates is not None and self.sd_probs[bi] > 0:
                    g1, g2 = (self._gates[bi, j, n0:n0 + N].contiguous() for j in (0, 1))
                bi += 1
                self._gemm_ln(att, P["proj" + nm], M, at.proj.bias, blk.norm1, xc, xc, x16, y, gate=g1, rows_per_gate=H * W)     # x = x + sd(norm1(attn(x)))
                hid = blk.mlp[0].out_features
                h16 = self._planes(f"h{s}", (M, hid))
                self._gemm(x16, P["fc1" + nm], M, bias=blk.mlp[0].bias, act_out=2, out16=h16)                      # GELU(Linear)
                self._gemm_ln(h16, P["fc2" + nm], M, blk.mlp[3].bias, blk.norm2, xc, xc, x16, y, gate=g2, rows_per_gate=H * W)                   # x = x + sd(norm2(mlp(x)))
            if merge is not None:
                Ho, Wo = (H + 1) // 2, (W + 1) // 2
                Mo = N * Ho * Wo
                m16 = self._planes(f"m{s}", (Mo, 4 * dim))
                ops.swin_merge16(xc.view(N, H, W, dim), m16[0], m16[1], prec)
                red = self._buf(f"red{s}", (Mo, 2 * dim))
                self._gemm(m16, P[f"red{s}"], Mo, out=red)
                H, W, M, dim = Ho, Wo, Mo, 2 * dim
                xc = self._buf(f"x{s + 1}", (M, dim))
                x16 = self._planes(f"x16.{s + 1}", (M, dim))
                ops.swin_ln(red, merge.norm.weight, merge.norm.bias, merge.norm.eps, None, xc, x16[0], x16[1], prec)
        xn = self._buf("xn", (M, dim))
        ops.swin_ln(xc, self.norm.weight, self.norm.bias, self.norm.eps, None, xn, None, None, prec)
        pooled = ops.swin_token_mean(xn.view(N, H * W, dim), self._buf("pool", (N, dim)))
        head = self.head
        ops.linear(pooled, ops.transpose(head.weight.detach().float().contiguous()), None if head.bias is None else head.bias.detach().float(), out)


_CONFIGS = {
    "swin_v2_t": dict(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), stochastic_depth_prob=0.2),
    "swin_v2_s": dict(embed_dim=96, depths=(2, 2, 18, 2), num_heads=(3, 6, 12, 24), stochastic_depth_prob=0.3),
    "swin_v2_b": dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), stochastic_depth_prob=0.5),
}


def get_model(name: str, **kwargs) -> SwinTransformerV2:
    """Stand-in for `torchvision.models.get_model(encoder)` (networks/s_zss_dm.py:19) for the Swin-V2 family."""
    if name not in _CONFIGS:
        raise NotImplementedError(f"HIP embedder: {name!r} is not built; available: {sorted(_CONFIGS)} (the reference's configs use swin_v2_t)")
    return SwinTransformerV2(**_CONFIGS[name], **kwargs)


def swin_v2_t(**kwargs) -> SwinTransformerV2:
    return get_model("swin_v2_t", **kwargs)
