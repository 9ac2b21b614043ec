"""ORACLE (test infrastructure, NOT product code): fp32 PyTorch-CPU restatement of the
reference's denoising U-Net, written from scratch in functional style over a flat
parameter dict that uses the reference's state-dict names.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product path (stedm_amd/) never does; it fails loudly without the HIP extension.

Pinned against golden vectors produced by importing the reference itself
(tests/golden/make_golden.py -> tests/golden/*.npz; checked by tests/test_oracle_golden.py).

Reference lines followed (all under /root/reference/):
  ldm/modules/diffusionmodules/util.py:151-171   timestep_embedding
  ldm/modules/diffusionmodules/util.py:199-216   normalization / GroupNorm32 (32 groups, eps 1e-5, fp32)
  ldm/modules/diffusionmodules/openaimodel.py:93-101   TimestepEmbedSequential routing
  ldm/modules/diffusionmodules/openaimodel.py:122-132  Upsample (nearest x2 then 3x3 conv)
  ldm/modules/diffusionmodules/openaimodel.py:156-173  Downsample (3x3 stride-2 pad-1 conv)
  ldm/modules/diffusionmodules/openaimodel.py:268-288  ResBlock._forward
  ldm/modules/diffusionmodules/openaimodel.py:291-297  ResBlockStyle
  ldm/modules/diffusionmodules/openaimodel.py:340-346, 378-394  AttentionBlock / QKVAttentionLegacy
  ldm/modules/diffusionmodules/openaimodel.py:465-739  UNetModel ctor (block structure)
  ldm/modules/diffusionmodules/openaimodel.py:761-806  UNetModel.forward
  ldm/modules/attention.py:37-64, 152-193, 196-215, 218-261  GEGLU/FeedForward, CrossAttention,
                                                            BasicTransformerBlock, SpatialTransformer
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


@dataclass
class UNetConfig:
    """Ctor kwargs of the reference UNetModel that the shipped configs exercise
    (openaimodel.py:465-492)."""
    image_size: int = 32
    in_channels: int = 7
    model_channels: int = 128
    out_channels: int = 4
    num_res_blocks: int = 2
    attention_resolutions: Sequence[int] = (32, 16, 8)
    channel_mult: Sequence[int] = (1, 4, 8)
    num_heads: int = 8
    num_head_channels: int = -1
    use_spatial_transformer: bool = False
    transformer_depth: int = 1
    context_dim: Optional[int] = None
    legacy: bool = True


def timestep_embedding(timesteps: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    """util.py:151-171 — cos half first, then sin; freq table built in fp32 on the host."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def group_norm32(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """util.py:214-216 — GroupNorm(32, C) evaluated in fp32."""
    return F.group_norm(x.float(), 32, w, b, eps).type(x.dtype)


def resblock(P: Params, pre: str, x: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    """openaimodel.py:268-288 with use_scale_shift_norm=False, dropout=0, no up/down."""
    h = group_norm32(x, P[pre + "in_layers.0.weight"], P[pre + "in_layers.0.bias"])
    h = F.conv2d(F.silu(h), P[pre + "in_layers.2.weight"], P[pre + "in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(emb), P[pre + "emb_layers.1.weight"], P[pre + "emb_layers.1.bias"])
    h = h + e[:, :, None, None]
    h = group_norm32(h, P[pre + "out_layers.0.weight"], P[pre + "out_layers.0.bias"])
    h = F.conv2d(F.silu(h), P[pre + "out_layers.3.weight"], P[pre + "out_layers.3.bias"], padding=1)
    if pre + "skip_connection.weight" in P:
        x = F.conv2d(x, P[pre + "skip_connection.weight"], P[pre + "skip_connection.bias"])
    return x + h


def qkv_attention_legacy(qkv: torch.Tensor, n_heads: int) -> torch.Tensor:
    """openaimodel.py:378-394 — head-major [h][q,k,v][ch] split, ch^-1/4 on q and k, fp32 softmax."""
    bs, width, length = qkv.shape
    ch = width // (3 * n_heads)
    q, k, v = qkv.reshape(bs * n_heads, ch * 3, length).split(ch, dim=1)
    scale = 1.0 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    w = torch.softmax(w.float(), dim=-1).type(w.dtype)
    a = torch.einsum("bts,bcs->bct", w, v)
    return a.reshape(bs, -1, length)


def attention_block(P: Params, pre: str, x: torch.Tensor, n_heads: int) -> torch.Tensor:
    """openaimodel.py:340-346."""
    b, c, *spatial = x.shape
    xf = x.reshape(b, c, -1)
    h = group_norm32(xf, P[pre + "norm.weight"], P[pre + "norm.bias"])
    qkv = F.conv1d(h, P[pre + "qkv.weight"], P[pre + "qkv.bias"])
    h = qkv_attention_legacy(qkv, n_heads)
    h = F.conv1d(h, P[pre + "proj_out.weight"], P[pre + "proj_out.bias"])
    return (xf + h).reshape(b, c, *spatial)


def cross_attention(P: Params, pre: str, x: torch.Tensor, context: Optional[torch.Tensor], heads: int) -> torch.Tensor:
    """attention.py:170-193 — context defaults to x; scale d_head^-1/2 after QK^T; softmax in input dtype."""
    ctx = x if context is None else context
    q = F.linear(x, P[pre + "to_q.weight"])
    k = F.linear(ctx, P[pre + "to_k.weight"])
    v = F.linear(ctx, P[pre + "to_v.weight"])
    b, n, inner = q.shape
    d = inner // heads

    def split(t):
        return t.reshape(t.shape[0], t.shape[1], heads, d).permute(0, 2, 1, 3).reshape(t.shape[0] * heads, t.shape[1], d)

    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum("bid,bjd->bij", q, k) * (d ** -0.5)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bij,bjd->bid", attn, v)
    out = out.reshape(b, heads, n, d).permute(0, 2, 1, 3).reshape(b, n, inner)
    return F.linear(out, P[pre + "to_out.0.weight"], P[pre + "to_out.0.bias"])


def basic_transformer_block(P: Params, pre: str, x: torch.Tensor, context, heads: int) -> torch.Tensor:
    """attention.py:211-215 (+ GEGLU feed-forward :37-64)."""
    dim = x.shape[-1]
    x = cross_attention(P, pre + "attn1.", F.layer_norm(x, (dim,), P[pre + "norm1.weight"], P[pre + "norm1.bias"]), None, heads) + x
    x = cross_attention(P, pre + "attn2.", F.layer_norm(x, (dim,), P[pre + "norm2.weight"], P[pre + "norm2.bias"]), context, heads) + x
    h = F.layer_norm(x, (dim,), P[pre + "norm3.weight"], P[pre + "norm3.bias"])
    h = F.linear(h, P[pre + "ff.net.0.proj.weight"], P[pre + "ff.net.0.proj.bias"])
    a, gate = h.chunk(2, dim=-1)
    h = a * F.gelu(gate)
    h = F.linear(h, P[pre + "ff.net.2.weight"], P[pre + "ff.net.2.bias"])
    return h + x


def spatial_transformer(P: Params, pre: str, x: torch.Tensor, context, heads: int, depth: int) -> torch.Tensor:
    """attention.py:250-261 — GroupNorm eps 1e-6 (:76-77)."""
    b, c, h, w = x.shape
    x_in = x
    y = F.group_norm(x, 32, P[pre + "norm.weight"], P[pre + "norm.bias"], 1e-6)
    y = F.conv2d(y, P[pre + "proj_in.weight"], P[pre + "proj_in.bias"])
    inner = y.shape[1]
    y = y.reshape(b, inner, h * w).permute(0, 2, 1)
    for d in range(depth):
        y = basic_transformer_block(P, f"{pre}transformer_blocks.{d}.", y, context, heads)
    y = y.permute(0, 2, 1).reshape(b, inner, h, w)
    y = F.conv2d(y, P[pre + "proj_out.weight"], P[pre + "proj_out.bias"])
    return y + x_in


# ----------------------------------------------------------------------------------------------
# Block structure (openaimodel.py:539-733), expressed as a flat op list so that the forward is a
# simple interpreter. Each op: (kind, state-dict prefix, meta).
# ----------------------------------------------------------------------------------------------
@dataclass
class Plan:
    input_blocks: List[List[Tuple[str, str, dict]]] = field(default_factory=list)
    middle: List[Tuple[str, str, dict]] = field(default_factory=list)
    output_blocks: List[List[Tuple[str, str, dict]]] = field(default_factory=list)
    shapes: Dict[str, Tuple[int, ...]] = field(default_factory=dict)  # state-dict name -> shape


def build_plan(cfg: UNetConfig) -> Plan:
    mc = cfg.model_channels
    ted = mc * 4
    plan = Plan()
    S = plan.shapes

    def conv(name, cin, cout, k):
        S[name + ".weight"] = (cout, cin, k, k)
        S[name + ".bias"] = (cout,)

    def norm(name, c):
        S[name + ".weight"] = (c,)
        S[name + ".bias"] = (c,)

    def lin(name, cin, cout, bias=True):
        S[name + ".weight"] = (cout, cin)
        if bias:
            S[name + ".bias"] = (cout,)

    def res(pre, cin, cout, emb_dim=ted):
        norm(pre + "in_layers.0", cin)
        conv(pre + "in_layers.2", cin, cout, 3)
        lin(pre + "emb_layers.1", emb_dim, cout)
        norm(pre + "out_layers.0", cout)
        conv(pre + "out_layers.3", cout, cout, 3)
        if cin != cout:
            conv(pre + "skip_connection", cin, cout, 1)

    lin("time_embed.0", mc, ted)
    lin("time_embed.2", ted, ted)
    conv("input_blocks.0.0", cfg.in_channels, mc, 3)
    plan.input_blocks.append([("conv", "input_blocks.0.0.", {})])
    chans = [mc]
    ch, ds, idx = mc, 1, 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            if ds in cfg.attention_resolutions:
                # openaimodel.py:580-590 calls layers.append() with no argument -> TypeError in the reference.
                raise TypeError("reference UNetModel cannot be built with ds in attention_resolutions (openaimodel.py:580)")
            pre = f"input_blocks.{idx}.0."
            res(pre, ch, mult * mc)
            plan.input_blocks.append([("res", pre, {})])
            ch = mult * mc
            chans.append(ch)
            idx += 1
        if level != len(cfg.channel_mult) - 1:
            pre = f"input_blocks.{idx}.0."
            conv(pre + "op", ch, ch, 3)
            plan.input_blocks.append([("down", pre, {})])
            chans.append(ch)
            ds *= 2
            idx += 1

    if cfg.num_head_channels == -1:
        heads = cfg.num_heads
    else:
        heads = ch // cfg.num_head_channels
    res("middle_block.0.", ch, ch)
    res("middle_block.1.block.", ch, ch, emb_dim=ted)  # style vector must be ted-wide (512)
    plan.middle.append(("res", "middle_block.0.", {}))
    plan.middle.append(("res_style", "middle_block.1.block.", {}))
    if not cfg.use_spatial_transformer:
        pre = "middle_block.2."
        norm(pre + "norm", ch)
        S[pre + "qkv.weight"] = (3 * ch, ch, 1)
        S[pre + "qkv.bias"] = (3 * ch,)
        S[pre + "proj_out.weight"] = (ch, ch, 1)
        S[pre + "proj_out.bias"] = (ch,)
        plan.middle.append(("attn", pre, {"heads": heads}))
    else:
        pre = "middle_block.2."
        d_head = ch // heads
        inner = heads * d_head
        cdim = cfg.context_dim
        norm(pre + "norm", ch)
        conv(pre + "proj_in", ch, inner, 1)
        for d in range(cfg.transformer_depth):
            tp = f"{pre}transformer_blocks.{d}."
            for a, kd in (("attn1", inner), ("attn2", cdim)):
                lin(tp + a + ".to_q", inner, inner, bias=False)
                lin(tp + a + ".to_k", kd, inner, bias=False)
                lin(tp + a + ".to_v", kd, inner, bias=False)
                lin(tp + a + ".to_out.0", inner, inner)
            lin(tp + "ff.net.0.proj", inner, inner * 8)
            lin(tp + "ff.net.2", inner * 4, inner)
            for n in ("norm1", "norm2", "norm3"):
                norm(tp + n, inner)
        conv(pre + "proj_out", inner, ch, 1)
        plan.middle.append(("st", pre, {"heads": heads, "depth": cfg.transformer_depth}))
    res("middle_block.3.", ch, ch)
    plan.middle.append(("res", "middle_block.3.", {}))

    oidx = 0
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            ops = []
            pre = f"output_blocks.{oidx}.0."
            res(pre, ch + ich, mc * mult)
            ops.append(("res", pre, {}))
            ch = mc * mult
            sub = 1
            if ds in cfg.attention_resolutions:
                # openaimodel.py:689-698: a second ResBlock(ch + ich -> ...) would be appended; its input
                # width no longer matches -> the reference crashes at run time. Not reachable with shipped configs.
                raise TypeError("reference UNetModel output block with ds in attention_resolutions is not runnable")
            if level and i == cfg.num_res_blocks:
                upre = f"output_blocks.{oidx}.{sub}."
                conv(upre + "conv", ch, ch, 3)
                ops.append(("up", upre, {}))
                ds //= 2
            plan.output_blocks.append(ops)
            oidx += 1
    norm("out.0", ch)
    conv("out.2", mc, cfg.out_channels, 3)
    return plan


def _run_ops(P: Params, ops, h, emb, context):
    for kind, pre, meta in ops:
        if kind == "conv":
            h = F.conv2d(h, P[pre + "weight"], P[pre + "bias"], padding=1)
        elif kind == "res":
            h = resblock(P, pre, h, emb)
        elif kind == "res_style":
            h = resblock(P, pre, h, context)  # openaimodel.py:97-98, 296-297: style vector IS the embedding
        elif kind == "down":
            h = F.conv2d(h, P[pre + "op.weight"], P[pre + "op.bias"], stride=2, padding=1)
        elif kind == "up":
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            h = F.conv2d(h, P[pre + "conv.weight"], P[pre + "conv.bias"], padding=1)
        elif kind == "attn":
            h = attention_block(P, pre, h, meta["heads"])
        elif kind == "st":
            h = spatial_transformer(P, pre, h, None, meta["heads"], meta["depth"])  # context never routed (openaimodel.py:99-100)
        else:
            raise ValueError(kind)
    return h


@torch.no_grad()
def unet_forward(P: Params, cfg: UNetConfig, x: torch.Tensor, timesteps: torch.Tensor,
                 context: Optional[torch.Tensor] = None, plan: Optional[Plan] = None,
                 taps: Optional[dict] = None) -> torch.Tensor:
    """openaimodel.py:761-806. `taps`, if given, receives per-block outputs for debugging/fixtures."""
    plan = plan or build_plan(cfg)
    t_emb = timestep_embedding(timesteps, cfg.model_channels)
    emb = F.linear(t_emb, P["time_embed.0.weight"], P["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), P["time_embed.2.weight"], P["time_embed.2.bias"])
    hs = []
    h = x.float()
    for i, ops in enumerate(plan.input_blocks):
        h = _run_ops(P, ops, h, emb, context)
        hs.append(h)
        if taps is not None:
            taps[f"in{i}"] = h
    h = _run_ops(P, plan.middle, h, emb, context)
    if taps is not None:
        taps["mid"] = h
    for i, ops in enumerate(plan.output_blocks):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_ops(P, ops, h, emb, context)
        if taps is not None:
            taps[f"out{i}"] = h
    h = group_norm32(h, P["out.0.weight"], P["out.0.bias"])
    return F.conv2d(F.silu(h), P["out.2.weight"], P["out.2.bias"], padding=1)
