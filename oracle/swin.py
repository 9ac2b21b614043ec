"""ORACLE (test infrastructure, NOT product code): fp32 PyTorch-CPU restatement of torchvision's Swin-Transformer-V2 classifier
(`swin_v2_t`), the style embedder of Agg_Mean / Agg_Max / Agg_Linear.

PARITY UNPINNED. torchvision is a third-party dependency of the reference (pinned `torchvision==0.18.1`, /root/reference/environment.yml:32),
absent from /root/reference and not installed in this image; the reference's own tests hold no vectors for it. This file restates the
published algorithm (Liu et al., "Swin Transformer V2: Scaling Up Capacity and Resolution", CVPR 2022, and the layer definitions of
torchvision.models.swin_transformer: SwinTransformer, SwinTransformerBlockV2, ShiftedWindowAttentionV2, PatchMergingV2,
shifted_window_attention) with explicit tensor operations (F.pad, torch.roll, window reshapes, a materialised shift mask), i.e. in a
different form than the product's index arithmetic. Anchors: the reference's call sites
  networks/s_zss_dm.py:19-20      embedder = torchvision.models.get_model("swin_v2_t"); embedder.head = Linear(768, 512)
  networks/agg_blocks.py:26-28    'b n h w c -> (b n) c h w' -> embedder -> [(b n), 512]

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Parameters: a flat dict with torchvision's state-dict names (features.0.0.weight, features.1.0.attn.qkv.weight, ..., norm.weight, head.weight).
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]
WS = 8


def relative_coords_table(ws: int = WS) -> torch.Tensor:
    """[(2ws-1)^2, 2] log-spaced relative coordinates: sign(8x/(ws-1)) log2(|8x/(ws-1)| + 1) / log2(8)."""
    c = torch.arange(-(ws - 1), ws, dtype=torch.float32)
    t = torch.stack(torch.meshgrid([c, c], indexing="ij"), dim=-1)    # [2ws-1, 2ws-1, 2] (dh, dw)
    t = t / (ws - 1) * 8
    t = torch.sign(t) * torch.log2(torch.abs(t) + 1.0) / 3.0
    return t.reshape(-1, 2)


def relative_position_index(ws: int = WS) -> torch.Tensor:
    """[ws^2, ws^2]: flat index of (dh + ws - 1, dw + ws - 1) for query i, key j."""
    ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
    ys, xs = ys.flatten(), xs.flatten()
    dh = ys[:, None] - ys[None, :] + ws - 1
    dw = xs[:, None] - xs[None, :] + ws - 1
    return dh * (2 * ws - 1) + dw


def position_bias(p: Params, pre: str, heads: int) -> torch.Tensor:
    """16 sigmoid(cpb_mlp(table))[index] -> [heads, ws^2, ws^2]. The table / index are recomputed (they are constants of the layer)."""
    t = relative_coords_table().to(p[pre + "cpb_mlp.0.weight"].dtype)
    h = F.relu(F.linear(t, p[pre + "cpb_mlp.0.weight"], p[pre + "cpb_mlp.0.bias"]))
    cpb = F.linear(h, p[pre + "cpb_mlp.2.weight"])                    # [(2ws-1)^2, heads]
    idx = relative_position_index()
    bias = cpb[idx.flatten()].view(WS * WS, WS * WS, heads).permute(2, 0, 1)
    return 16 * torch.sigmoid(bias)


def window_attention(x: torch.Tensor, p: Params, pre: str, heads: int, shift: int) -> torch.Tensor:
    """shifted_window_attention + ShiftedWindowAttentionV2.forward: x [B, H, W, C] -> [B, H, W, C]."""
    B, H, W, C = x.shape
    pad_r, pad_b = (WS - W % WS) % WS, (WS - H % WS) % WS
    x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
    pH, pW = H + pad_b, W + pad_r
    sh = 0 if WS >= pH else shift      # no shift along a side the window covers
    sw = 0 if WS >= pW else shift
    if sh + sw > 0:
        x = torch.roll(x, shifts=(-sh, -sw), dims=(1, 2))
    nW = (pH // WS) * (pW // WS)
    x = x.view(B, pH // WS, WS, pW // WS, WS, C).permute(0, 1, 3, 2, 4, 5).reshape(B * nW, WS * WS, C)
    qkv_b = p[pre + "qkv.bias"].clone()
    n = qkv_b.numel() // 3
    qkv_b[n:2 * n].zero_()
    qkv = F.linear(x, p[pre + "qkv.weight"], qkv_b).reshape(B * nW, WS * WS, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1)
    attn = attn * torch.clamp(p[pre + "logit_scale"], max=math.log(100.0)).exp()
    attn = attn + position_bias(p, pre, heads).unsqueeze(0)
    if sh + sw > 0:
        m = x.new_zeros((pH, pW))
        hs = ((0, -WS), (-WS, -sh), (-sh, None))
        wsl = ((0, -WS), (-WS, -sw), (-sw, None))
        cnt = 0
        for a in hs:
            for b in wsl:
                m[a[0]:a[1], b[0]:b[1]] = cnt
                cnt += 1
        m = m.view(pH // WS, WS, pW // WS, WS).permute(0, 2, 1, 3).reshape(nW, WS * WS)
        m = m.unsqueeze(1) - m.unsqueeze(2)
        m = m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)
        attn = attn.view(B, nW, heads, WS * WS, WS * WS) + m.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, heads, WS * WS, WS * WS)
    attn = F.softmax(attn, dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B * nW, WS * WS, C)
    x = F.linear(x, p[pre + "proj.weight"], p[pre + "proj.bias"])
    x = x.view(B, pH // WS, pW // WS, WS, WS, C).permute(0, 1, 3, 2, 4, 5).reshape(B, pH, pW, C)
    if sh + sw > 0:
        x = torch.roll(x, shifts=(sh, sw), dims=(1, 2))
    return x[:, :H, :W, :].contiguous()


def block(x: torch.Tensor, p: Params, pre: str, heads: int, shift: int, gates=None) -> torch.Tensor:
    """SwinTransformerBlockV2.forward. gates [2, N] (train mode): torchvision.ops.StochasticDepth(p, "row") on both residual branches, i.e.
    the branch of image n times gates[j, n] = bernoulli(1 - p) / (1 - p); None (eval): identity. Dropout is 0 in swin_v2_*."""
    C = x.shape[-1]
    a = F.layer_norm(window_attention(x, p, pre + "attn.", heads, shift), (C,), p[pre + "norm1.weight"], p[pre + "norm1.bias"], 1e-5)
    x = x + (a if gates is None else a * gates[0].view(-1, 1, 1, 1))
    h = F.gelu(F.linear(x, p[pre + "mlp.0.weight"], p[pre + "mlp.0.bias"]))
    h = F.linear(h, p[pre + "mlp.3.weight"], p[pre + "mlp.3.bias"])
    m = F.layer_norm(h, (C,), p[pre + "norm2.weight"], p[pre + "norm2.bias"], 1e-5)
    return x + (m if gates is None else m * gates[1].view(-1, 1, 1, 1))


def patch_merging(x: torch.Tensor, p: Params, pre: str) -> torch.Tensor:
    """PatchMergingV2.forward: pad to even sides, concat the 2x2 neighbours, Linear(4C, 2C, no bias), LayerNorm(2C)."""
    H, W = x.shape[1:3]
    x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, 0::2, 0::2, :], x[:, 1::2, 0::2, :], x[:, 0::2, 1::2, :], x[:, 1::2, 1::2, :]], -1)
    x = F.linear(x, p[pre + "reduction.weight"])
    return F.layer_norm(x, (x.shape[-1],), p[pre + "norm.weight"], p[pre + "norm.bias"], 1e-5)


def swin_v2_forward(p: Params, x: torch.Tensor, depths: List[int] = (2, 2, 6, 2), num_heads: List[int] = (3, 6, 12, 24), sd_gates=None) -> torch.Tensor:
    """SwinTransformer.forward: x [N, 3, H, W] -> [N, classes]. sd_gates [blocks, 2, N]: train-mode stochastic-depth gates (see block)."""
    x = F.conv2d(x, p["features.0.0.weight"], p["features.0.0.bias"], stride=4).permute(0, 2, 3, 1)
    x = F.layer_norm(x, (x.shape[-1],), p["features.0.2.weight"], p["features.0.2.bias"], 1e-5)
    fi, bi = 1, 0
    for s, depth in enumerate(depths):
        for i in range(depth):
            x = block(x, p, f"features.{fi}.{i}.", num_heads[s], 0 if i % 2 == 0 else WS // 2, None if sd_gates is None else sd_gates[bi])
            bi += 1
        fi += 1
        if s < len(depths) - 1:
            x = patch_merging(x, p, f"features.{fi}.")
            fi += 1
    x = F.layer_norm(x, (x.shape[-1],), p["norm.weight"], p["norm.bias"], 1e-5)
    x = x.permute(0, 3, 1, 2).mean(dim=(2, 3))
    return F.linear(x, p["head.weight"], p["head.bias"])
