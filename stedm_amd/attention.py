"""SpatialTransformer (ldm/modules/attention.py:218-261) with the reference's parameter names, HIP-backed.

As wired by the reference U-Net (TimestepEmbedSequential calls `layer(x)`, openaimodel.py:99-100) the block never receives
a context: both attentions of a BasicTransformerBlock are self-attentions, and `attn2.to_k/to_v` (Linear(context_dim, inner))
only accept the tokens when context_dim == inner_dim. The same restriction is enforced here.

Kernels reused: gn_apply16 (GroupNorm eps 1e-6), ln_apply16, the register-streamed 1x1 GEMM, the attention kernels of the AttentionBlock
(the per-head q|k|v channel layout is produced by permuting the packed to_q/to_k/to_v rows once; scale d^-1/2 == (d^-1/4 on q) *
(d^-1/4 on k)): attn_flash_kernel on MFMA in the single-product modes (stedm_attn_legacy16), the fp32 kernel in the 3-product modes
and under training; geglu16.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn as nn

from . import ops


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.):
        super().__init__()
        assert glu, "BasicTransformerBlock uses gated_ff=True (attention.py:197,200)"
        inner_dim = int(dim * mult)
        self.net = nn.Sequential(GEGLU(dim, inner_dim), nn.Dropout(dropout), nn.Linear(inner_dim, dim_out or dim))


class CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        inner_dim = dim_head * heads
        context_dim = context_dim if context_dim is not None else query_dim
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        self.to_q = nn.Linear(query_dim, inner_dim, bias=False)
        self.to_k = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_v = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, query_dim), nn.Dropout(dropout))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, n_heads, d_head, dropout=0., context_dim=None, gated_ff=True, checkpoint=True):
        super().__init__()
        self.attn1 = CrossAttention(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = CrossAttention(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)


class SpatialTransformer(nn.Module):
    """attention.py:218-261 (container + HIP runner `run(x_nhwc, engine)`)."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0., context_dim=None):
        super().__init__()
        self.in_channels = in_channels
        self.n_heads, self.d_head = n_heads, d_head
        self.context_dim = context_dim
        inner_dim = n_heads * d_head
        self.inner_dim = inner_dim
        self.norm = nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner_dim, kernel_size=1, stride=1, padding=0)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner_dim, n_heads, d_head, dropout=dropout, context_dim=context_dim)
                                                 for _ in range(depth)])
        self.proj_out = zero_module(nn.Conv2d(inner_dim, in_channels, kernel_size=1, stride=1, padding=0))

    # ---- weight packing (called from UNetModel._prepare)
    def pack(self, prec) -> Dict:
        P = {}

        def cv(w):
            """([cout][1][cin] hi, lo planes; fragment-order pack of the register-streamed 1x1 kind (single-product modes: 32x32x16 order,
            3-product modes: the hi + lo 16x16x32 streams) or None when the width does not admit it)"""
            w4 = w.detach().float().reshape(w.shape[0], -1, 1, 1).contiguous()
            hi, lo = ops.pack_conv_weight(w4, prec)
            cin = w4.shape[1]
            frag = ops.pack_conv_weight_frag(w4, prec) if prec.npass == 1 and cin % 64 == 0 else None
            frag16 = ops.pack_conv_weight_frag16(w4, prec) if prec.npass == 3 and cin % 32 == 0 and cin >= 64 else None
            return hi, lo, frag, frag16
        P["proj_in"] = cv(self.proj_in.weight)
        P["proj_out"] = cv(self.proj_out.weight)
        h, d = self.n_heads, self.d_head
        for i, blk in enumerate(self.transformer_blocks):
            for nm in ("attn1", "attn2"):
                at = getattr(blk, nm)
                if at.to_k.in_features != self.inner_dim:
                    # attn2.to_k is Linear(context_dim, inner) but is fed the tokens (width inner): the reference fails in
                    # torch.nn.functional.linear here (SURVEY.md §8a A8 [probe])
                    raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied (tokens x {self.inner_dim} and "
                                       f"{at.to_k.in_features} x {self.inner_dim}): SpatialTransformer without context needs "
                                       "context_dim == n_heads * d_head")
                # legacy layout rows: [head][q | k | v][d]
                w = torch.stack([at.to_q.weight.detach().float().reshape(h, d, -1), at.to_k.weight.detach().float().reshape(h, d, -1),
                                 at.to_v.weight.detach().float().reshape(h, d, -1)], dim=1).reshape(3 * h * d, -1)
                P[f"{i}.{nm}.qkv"] = cv(w)
                P[f"{i}.{nm}.out"] = cv(at.to_out[0].weight)
            P[f"{i}.ff.proj"] = cv(blk.ff.net[0].proj.weight)
            P[f"{i}.ff.out"] = cv(blk.ff.net[2].weight)
        return P

    @torch.no_grad()
    def run(self, x, P, prec, buf, save=None):
        """x [B,H,W,C] NHWC fp32 -> same shape. `buf(name, shape, dtype)` is the owning model's buffer cache.
        save (a dict, training): every stage keeps its own tensors — the token stream y0 (after proj_in), y{i}a / y{i}b / y{i}c (after attn1,
        attn2 and the feed-forward of block i), the qkv rows and attention outputs of both attentions, the GEGLU input g — and they are
        recorded in it for UNetTrainer's backward; without it the residual GEMMs update one token stream in place."""
        B, H, W, C = x.shape
        T, inner = H * W, self.inner_dim
        M = B * T
        i16 = torch.int16
        lo_ok = prec.npass == 3
        planes = lambda nm, width: (buf(f"st.{nm}.hi.{M}x{width}", (M, width), i16), buf(f"st.{nm}.lo.{M}x{width}", (M, width), i16) if lo_ok else None)
        v4 = lambda t, width: None if t is None else t.view(1, 1, M, width)
        tag = f"st{id(self)}." if save is not None else "st."

        def gemm(a16, w, N, bias=None, res=None, out=None, out16=None):
            K = a16[0].shape[-1]
            ops.conv_igemm(None, w[0], w[1], v4(out, N), prec=prec, ks=1, src16=(v4(a16[0], K), v4(a16[1], K)), bias=bias, res=v4(res, N),
                           w_frag=w[2], w_frag16=w[3], out16=None if out16 is None else (v4(out16, N), None))

        # single-product modes, inference: CrossAttention.forward (attention.py:170-193) on MFMA - the stacked to_q | to_k | to_v GEMM writes
        # its rows as ONE 16-bit plane (no fp32 qkv), attn_flash_kernel reads K / V tiles from it and writes to_out's operand plane
        # (same operand rounding as every other contraction of these modes; logits, softmax and normalisation fp32). The 3-product modes
        # and the training forward (the backward re-reads fp32 qkv rows) keep the fp32 kernel.
        mfma_attn = prec.npass == 1 and save is None and self.d_head in (16, 32, 64, 128)

        # GroupNorm (eps 1e-6, no activation) -> proj_in
        g16 = planes("gn", C)
        stats = buf("st.gnpart", (B * ops.gn_nslab(C, T) * 32 * 2,), torch.float64)
        ops.gn_stats(x, None, stats, 32)
        ops.gn_apply16(x, None, g16[0].view(B, H, W, C), None if g16[1] is None else g16[1].view(B, H, W, C), prec, self.norm.weight,
                       self.norm.bias, self.norm.eps, 32, 0, stats)
        y = buf(f"{tag}y.{M}x{inner}", (M, inner))
        gemm(g16, P["proj_in"], inner, bias=self.proj_in.bias, out=y)
        if save is not None:
            save["y0"] = y
        ln = planes("ln", inner)
        a16 = planes("a", inner)
        for i, blk in enumerate(self.transformer_blocks):
            for nm, norm in (("attn1", blk.norm1), ("attn2", blk.norm2)):
                at = getattr(blk, nm)
                sfx = f"{i}.{nm}.{M}" if save is not None else f"{M}"
                ops.ln_apply16(y, norm.weight, norm.bias, norm.eps, ln[0], ln[1], prec)
                if mfma_attn:
                    qkv16 = buf(f"st.qkv16.{M}x{3 * inner}", (B, T, 3 * inner), i16)
                    gemm(ln, P[f"{i}.{nm}.qkv"], 3 * inner, out16=qkv16)
                    ops.attn_legacy16(qkv16, a16[0], self.n_heads, prec)
                    gemm(a16, P[f"{i}.{nm}.out"], inner, bias=at.to_out[0].bias, res=y, out=y)
                    continue
                qkv = buf(f"{tag}qkv.{sfx}", (B, T, 3 * inner))
                att = buf(f"{tag}att.{sfx}", (B, T, inner))
                gemm(ln, P[f"{i}.{nm}.qkv"], 3 * inner, out=qkv)
                ops.attn_legacy(qkv, att, self.n_heads)
                ops.gn_apply16(att.view(1, 1, M, inner), None, a16[0].view(1, 1, M, inner), None if a16[1] is None else a16[1].view(1, 1, M, inner), prec)
                yn = buf(f"{tag}y.{i}.{nm}.{M}x{inner}", (M, inner)) if save is not None else y
                gemm(a16, P[f"{i}.{nm}.out"], inner, bias=at.to_out[0].bias, res=y, out=yn)
                if save is not None:
                    save[f"{i}.{nm}"] = (y, qkv, att, yn)          # (input tokens, qkv rows, attention output, output tokens)
                y = yn
            ops.ln_apply16(y, blk.norm3.weight, blk.norm3.bias, blk.norm3.eps, ln[0], ln[1], prec)
            g = buf(f"{tag}ffg.{i}.{M}" if save is not None else f"st.ffg.{M}", (M, 8 * inner))
            gemm(ln, P[f"{i}.ff.proj"], 8 * inner, bias=blk.ff.net[0].proj.bias, out=g)
            h16 = planes("ffh", 4 * inner)
            ops.geglu16(g, h16[0], h16[1], prec)
            yn = buf(f"{tag}y.{i}.ff.{M}x{inner}", (M, inner)) if save is not None else y
            gemm(h16, P[f"{i}.ff.out"], inner, bias=blk.ff.net[2].bias, res=y, out=yn)
            if save is not None:
                save[f"{i}.ff"] = (y, g, yn)                       # (input tokens, GEGLU input, output tokens)
            y = yn
        out = buf(f"{tag}out.{B}x{H}x{W}x{C}", (B, H, W, C))
        ops.gn_apply16(y.view(1, 1, M, inner), None, a16[0].view(1, 1, M, inner), None if a16[1] is None else a16[1].view(1, 1, M, inner), prec)
        ops.conv_igemm(None, P["proj_out"][0], P["proj_out"][1], out.view(1, 1, M, C), prec=prec, ks=1,
                       src16=(a16[0].view(1, 1, M, inner), None if a16[1] is None else a16[1].view(1, 1, M, inner)),
                       bias=self.proj_out.bias, res=x.view(1, 1, M, C), w_frag=P["proj_out"][2], w_frag16=P["proj_out"][3])
        if save is not None:
            save["yL"] = y
        return out
