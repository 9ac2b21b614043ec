"""First stage of the latent diffusion model: the frozen VQ-f4 autoencoder, HIP-backed.

Drop-in for `ldm.models.autoencoder.VQModelInterface` (reference autoencoder.py:264-282 over VQModel :14-110) with
`ldm.modules.diffusionmodules.model.Encoder / Decoder` (model.py:368-459, 462-568: ResnetBlock :82-140, AttnBlock :143-199,
Upsample :43-57, Downsample :59-79, GroupNorm eps 1e-6 :38-39) and taming-transformers' `VectorQuantizer2` (un-vendored third-party
dependency, autoencoder.py:6; its published forward is restated: nearest codebook entry + straight-through value). Same constructor
arguments (`embed_dim, n_embed, ddconfig, lossconfig, ckpt_path, ...`), same `encode(x)` / `decode(h, force_not_quantize=False)`,
same state-dict names (`encoder.down.0.block.0.norm1.weight`, `decoder.up.2.upsample.conv.weight`, `quantize.embedding.weight`,
`quant_conv.*`, `post_quant_conv.*`), so the reference's `vq-f4.ckpt` loads with `load_state_dict`.

torch.nn modules are parameter containers; the arithmetic runs in the kernels the U-Net uses: GroupNorm(+SiLU) from producer-side
channel statistics into 16-bit operand planes (stedm_gn_apply16c), 3x3 / 1x1 / fused-shortcut / sub-pixel-upsample /
space-to-depth-downsample convolutions on MFMA (stedm_conv_igemm), the boundary convs (stedm_conv_in / stedm_conv_out), and for the
single-head attention of width C two GEMMs around a row softmax (stedm_softmax_rows16). Activations are NHWC fp32 inside, NCHW at the
surface like the reference."""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from ._lib import CONV_S2D, CONV_UP_SUBPIXEL, StedmHipError
from .ops import Precision


def Normalize(in_channels, num_groups=32):
    """model.py:38-39."""
    return nn.GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


class Upsample(nn.Module):
    """model.py:43-57 (container)."""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if not with_conv:
            raise NotImplementedError("Upsample without conv (resamp_with_conv=False) is not used by vq-f4.yaml")
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)


class Downsample(nn.Module):
    """model.py:59-79 (container): F.pad(x, (0,1,0,1)) then conv 3x3 stride 2 padding 0."""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if not with_conv:
            raise NotImplementedError("Downsample without conv (resamp_with_conv=False) is not used by vq-f4.yaml")
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)


class ResnetBlock(nn.Module):
    """model.py:82-140 (container; temb_channels = 0 in Encoder / Decoder, so there is no temb_proj)."""

    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        if conv_shortcut or temb_channels > 0 or dropout != 0.0:
            raise NotImplementedError("ResnetBlock: conv_shortcut / temb / dropout are not used by the first stage (vq-f4.yaml)")
        self.in_channels = in_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.out_channels = out_channels
        self.use_conv_shortcut = conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if self.in_channels != self.out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)


class AttnBlock(nn.Module):
    """model.py:143-199 (container)."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.k = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.v = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.proj_out = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)


def make_attn(in_channels, attn_type="vanilla"):
    assert attn_type in ["vanilla", "linear", "none"], f'attn_type {attn_type} unknown'
    if attn_type == "vanilla":
        return AttnBlock(in_channels)
    if attn_type == "none":
        return nn.Identity(in_channels)
    raise NotImplementedError("LinAttnBlock is not used by vq-f4.yaml")


class Encoder(nn.Module):
    """model.py:368-459 (container)."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0, resamp_with_conv=True, in_channels,
                 resolution, z_channels, double_z=True, use_linear_attn=False, attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        if use_linear_attn:
            attn_type = "linear"
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution = resolution
        self.in_channels = in_channels
        self.conv_in = nn.Conv2d(in_channels, self.ch, kernel_size=3, stride=1, padding=1)
        curr_res = resolution
        in_ch_mult = (1,) + tuple(ch_mult)
        self.in_ch_mult = in_ch_mult
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in = ch * in_ch_mult[i_level]
            block_out = ch * ch_mult[i_level]
            for _ in range(self.num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=self.temb_ch, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(make_attn(block_in, attn_type=attn_type))
            down = nn.Module()
            down.block, down.attn = block, attn
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
                curr_res = curr_res // 2
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, kernel_size=3, stride=1, padding=1)


class Decoder(nn.Module):
    """model.py:462-568 (container)."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0, resamp_with_conv=True, in_channels,
                 resolution, z_channels, give_pre_end=False, tanh_out=False, use_linear_attn=False, attn_type="vanilla", **ignorekwargs):
        super().__init__()
        if use_linear_attn:
            attn_type = "linear"
        if give_pre_end or tanh_out:
            raise NotImplementedError("Decoder: give_pre_end / tanh_out are not used by vq-f4.yaml")
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution = resolution
        self.in_channels = in_channels
        self.give_pre_end, self.tanh_out = give_pre_end, tanh_out
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)
        self.conv_in = nn.Conv2d(z_channels, block_in, kernel_size=3, stride=1, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(self.num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=self.temb_ch, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(make_attn(block_in, attn_type=attn_type))
            up = nn.Module()
            up.block, up.attn = block, attn
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
                curr_res = curr_res * 2
            self.up.insert(0, up)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, kernel_size=3, stride=1, padding=1)


class VectorQuantizer(nn.Module):
    """taming.modules.vqvae.quantize.VectorQuantizer2 (third-party, not under /root/reference; call sites autoencoder.py:39-41, 277):
    codebook `embedding` [n_e, e_dim]; forward(z) -> (z_q, loss, (perplexity, min_encodings, min_encoding_indices)). The eval path
    computes d = |z|^2 + |e|^2 - 2 z.e per latent pixel, takes argmin, looks the entry up and returns z + (z_q - z).detach()."""

    def __init__(self, n_e, e_dim, beta=0.25, remap=None, unknown_index="random", sane_index_shape=False, legacy=True):
        super().__init__()
        if remap is not None:
            raise NotImplementedError("VectorQuantizer remap is not used by vq-f4.yaml")
        self.n_e, self.e_dim, self.beta, self.legacy = n_e, e_dim, beta, legacy
        self.sane_index_shape = sane_index_shape
        self.embedding = nn.Embedding(self.n_e, self.e_dim)
        self.embedding.weight.data.uniform_(-1.0 / self.n_e, 1.0 / self.n_e)

    @torch.no_grad()
    def forward(self, z, temp=None, rescale_logits=False, return_logits=False):
        idx, z_q = ops.vq_nearest(z.float().contiguous(), self.embedding.weight.detach().float().contiguous())
        ind = idx if self.sane_index_shape else idx.reshape(-1)
        return z_q, None, (None, None, ind)

    @torch.no_grad()
    def get_codebook_entry(self, indices, shape):
        z_q = self.embedding.weight[indices.reshape(-1)]
        if shape is not None:
            z_q = z_q.view(shape).permute(0, 3, 1, 2).contiguous()
        return z_q


class _Packed:
    __slots__ = ("hi", "lo", "bias", "frag", "extra", "frag16", "extra_src")

    def __init__(self, hi, lo, bias, frag=None, extra=None, frag16=None, extra_src=None):
        self.hi, self.lo, self.bias, self.frag, self.extra, self.frag16, self.extra_src = hi, lo, bias, frag, extra, frag16, extra_src


class VQModelInterface(nn.Module):
    """autoencoder.py:264-282 over VQModel (:14-110). Extra (non-reference) argument: `precision` (ops.Precision name; the parity mode
    is the one asserted against the oracle, bf16 / f16 are the single-product fast modes)."""

    def __init__(self, embed_dim, ddconfig=None, lossconfig=None, n_embed=None, ckpt_path=None, ignore_keys=(), image_key="image",
                 colorize_nlabels=None, monitor=None, batch_resize_range=None, scheduler_config=None, lr_g_factor=1.0, remap=None,
                 sane_index_shape=False, use_ema=False, precision: str = "parity"):
        super().__init__()
        if ddconfig is None or n_embed is None:
            raise TypeError("VQModelInterface needs ddconfig and n_embed (conf/diffusion/first_stage_config/vq-f4.yaml)")
        if use_ema or colorize_nlabels is not None or batch_resize_range is not None:
            raise NotImplementedError("use_ema / colorize_nlabels / batch_resize_range: training-side options of VQModel, unused by the frozen stage")
        self.embed_dim, self.n_embed, self.image_key = embed_dim, n_embed, image_key
        dd = dict(ddconfig)
        self.encoder = Encoder(**dd)
        self.decoder = Decoder(**dd)
        self.loss = nn.Identity()                     # lossconfig: torch.nn.Identity (vq-f4.yaml:22-23)
        self.quantize = VectorQuantizer(n_embed, embed_dim, beta=0.25, remap=remap, sane_index_shape=sane_index_shape)
        self.quant_conv = nn.Conv2d(dd["z_channels"], embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, dd["z_channels"], 1)
        if monitor is not None:
            self.monitor = monitor
        self.precision = Precision.parse(precision) if isinstance(precision, str) else precision
        self._packed: Dict = {}
        self._pack_key = None
        self._bufs: Dict[Tuple, torch.Tensor] = {}
        self._cs: Dict[int, torch.Tensor] = {}
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)
        for p in self.parameters():                   # instantiate_first_stage freezes it (ddpm.py:530-535)
            p.requires_grad = False

    def init_from_ckpt(self, path, ignore_keys=()):
        """autoencoder.py:78-90."""
        # a third-party checkpoint (vq-f4.ckpt): tensors only, never a full unpickle — if the safe loader refuses the file, say so and stop
        try:
            ck = torch.load(path, map_location="cpu", weights_only=True)
        except Exception as e:
            raise RuntimeError(f"{path}: torch.load(weights_only=True) refused this checkpoint ({type(e).__name__}: {e}); it is not loaded "
                               f"with a full unpickle — re-save its state_dict as plain tensors (e.g. safetensors)") from e
        sd = ck["state_dict"] if isinstance(ck, dict) and "state_dict" in ck else ck
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                del sd[k]
        missing, unexpected = self.load_state_dict(sd, strict=False)
        print(f"Restored from {path} with {len(missing)} missing and {len(unexpected)} unexpected keys")

    def set_precision(self, precision) -> None:
        self.precision = Precision.parse(precision) if isinstance(precision, str) else precision
        self._packed.clear()
        self._pack_key = None

    # ------------------------------------------------------------------------------------------------ engine plumbing
    def _buf(self, name, shape, dtype=torch.float32):
        key = (name, tuple(shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            t = torch.empty(tuple(shape), dtype=dtype, device=self.quant_conv.weight.device)
            self._bufs[key] = t
        return t

    def _planes(self, shape, kind="a16"):
        hi = self._buf(f"{kind}.hi", shape, torch.int16)
        lo = self._buf(f"{kind}.lo", shape, torch.int16) if self.precision.npass == 3 else None
        return hi, lo

    def _cs_new(self, t, nslab=None):
        B, C = t.shape[0], t.shape[-1]
        HW = t.numel() // (B * C)
        cs = self._buf(f"cs.{t.data_ptr()}", (B, nslab or ops.gn_chan_nslab(HW), C, 2))
        self._cs[t.data_ptr()] = cs
        return cs

    def _chan_stats(self, t):
        cs = self._cs.get(t.data_ptr())
        if cs is None:
            cs = self._cs_new(t)
            ops.gn_chan_stats(t, cs)
        return cs

    def _norm16(self, norm: Optional[nn.GroupNorm], act: int, x, want_raw=False, kind="a16"):
        """act(GroupNorm(x)) (norm None: plain conversion) as 16-bit operand planes; want_raw: also the plain conversion of x (operand of
        the 1x1 nin_shortcut), from the same pass."""
        hi, lo = self._planes(tuple(x.shape), kind)
        if norm is None:
            ops.gn_apply16(x, None, hi, lo, self.precision)
            return hi, lo
        raw = self._planes(tuple(x.shape), "raw16") if want_raw else None
        ops.gn_apply16c(x, self._chan_stats(x), None, None, hi, lo, self.precision, norm.weight, norm.bias, norm.eps, norm.num_groups, act, 0, raw)
        return ((hi, lo), raw) if want_raw else (hi, lo)

    def _ws(self, nel):
        if nel > (1 << 23):
            return None
        return self._buf("conv_ws", ((16 if nel <= (1 << 20) else (4 if nel <= (1 << 22) else 2)) * nel,))

    def _prepare(self):
        params = list(self.parameters())
        dev = params[0].device
        if dev.type != "cuda":
            raise StedmHipError("VQModelInterface needs its parameters on the GPU; there is no CPU fallback")
        key = (self.precision, dev, tuple(p._version for p in params), tuple(p.data_ptr() for p in params))
        if key == self._pack_key:
            return
        self._packed.clear()
        prec = self.precision
        one = Precision(prec.mm_dtype, 1)

        def pack(conv):
            w4 = conv.weight.detach().float().contiguous()
            ks = w4.shape[-1]
            hi = lo = frag = None
            ok = (ks == 3 and conv.in_channels % 16 == 0) or (ks == 1 and conv.in_channels % 64 == 0)
            frag16 = None
            if prec.npass == 1 and ok and conv.stride == (1, 1):
                # both fragment orders (the stage is frozen: packed once): the 16x16x32 MFMA kind takes a 3x3 from 256 input channels on
                # unless its two-plane chunk ring does not fit LDS (rows of 256+ pixels), where the 32x32x16 kind runs
                frag = ops.pack_conv_weight_frag(w4, prec)
                if conv.in_channels % 32 == 0:
                    frag16 = ops.pack_conv_weight_frag16(w4, prec)
                hi = ops.LazyPlanes(lambda w=w4: ops.pack_conv_weight(w, prec))
            else:
                hi, lo = ops.pack_conv_weight(w4, prec)
            self._packed[id(conv)] = _Packed(hi, lo, None if conv.bias is None else conv.bias.detach().float().contiguous(), frag, None, frag16,
                                             extra_src=w4 if (prec.npass == 3 and ks == 3) else None)

        for m in self.modules():
            if isinstance(m, ResnetBlock):
                pack(m.conv1); pack(m.conv2)
                if m.in_channels != m.out_channels:
                    pack(m.nin_shortcut)
            elif isinstance(m, AttnBlock):
                for c in (m.q, m.k, m.v, m.proj_out):
                    pack(c)
            elif isinstance(m, Upsample):
                w = m.conv.weight.detach().float().contiguous()
                hi, lo = ops.pack_conv_weight_up(w, prec)
                frag = ops.pack_conv_weight_up_frag(w, prec) if prec.npass == 1 and m.conv.in_channels % 32 == 0 else None
                self._packed[id(m.conv)] = _Packed(hi, lo, m.conv.bias.detach().float().contiguous(), frag)
                pack(m.conv)                                   # the plain 3x3 form, for inputs too wide for the sub-pixel kernel's LDS ring
                self._packed[("plain", id(m.conv))] = self._packed.pop(id(m.conv))
                self._packed[id(m.conv)] = _Packed(hi, lo, m.conv.bias.detach().float().contiguous(), frag)
            elif isinstance(m, Downsample):
                # bottom/right-padded stride-2 conv as a 2x2 conv over space-to-depth planes (register-streamed kernel, single product);
                # the 3-product parity mode runs it as hi*hi + lo*hi + hi*lo with the residual weights packed beside
                w = m.conv.weight.detach().float().contiguous()
                frag = ops.pack_conv_weight_s2d_frag(w, one, pad_br=True)
                extra = None
                if prec.npass == 3:
                    wh = w.to(torch.float16 if prec.mm_dtype == 0 else torch.bfloat16).float()
                    extra = ops.pack_conv_weight_s2d_frag((w - wh).contiguous(), one, pad_br=True)
                self._packed[id(m.conv)] = _Packed(None, None, m.conv.bias.detach().float().contiguous(), frag, extra)
        ci = self.decoder.conv_in
        wpad = torch.zeros((ci.out_channels, 32, 3, 3), dtype=torch.float32, device=dev)
        wpad[:, :ci.in_channels].copy_(ci.weight.detach().float())
        hi, lo = ops.pack_conv_weight(wpad, prec)
        self._packed[id(ci)] = _Packed(hi, lo, ci.bias.detach().float().contiguous(),
                                       ops.pack_conv_weight_frag(wpad, prec) if prec.npass == 1 else None)
        self._out_w = {id(c): ops.conv_out_weight(c.weight) for c in (self.encoder.conv_out, self.decoder.conv_out)}
        self._pack_key = key

    # ------------------------------------------------------------------------------------------------ block runners (NHWC fp32)
    def _wide3(self, pk, src16, out, res, stats):
        """3x3 on rows of 256+ pixels in the 3-product parity mode: only the register-streamed kernel tiles such rows and it is a
        single-product kernel, so the three products run as three launches over the hi / lo planes (lo*hi + hi*lo + hi*hi, fp32
        accumulation through `res`), with the weight's hi part and residual packed in fragment order on first use"""
        prec = self.precision
        one = Precision(prec.mm_dtype, 1)
        if pk.extra is None:
            w = pk.extra_src
            wh = w.to(torch.float16 if prec.mm_dtype == 0 else torch.bfloat16).float()
            pk.extra = (ops.pack_conv_weight_frag(wh, one), ops.pack_conv_weight_frag((w - wh).contiguous(), one))
        fh, fl = pk.extra
        lazy = lambda: ops.LazyPlanes(lambda: (_ for _ in ()).throw(StedmHipError("wide 3-product conv: the register-streamed kernel refused the shape")))
        ws = self._ws(out.numel())
        ops.conv_igemm(None, lazy(), None, out, prec=one, src16=(src16[1], None), w_frag=fh, res=res, ws=ws)
        ops.conv_igemm(None, lazy(), None, out, prec=one, src16=(src16[0], None), w_frag=fl, res=out, ws=ws)
        return ops.conv_igemm(None, lazy(), None, out, prec=one, src16=(src16[0], None), w_frag=fh, bias=pk.bias, res=out, ws=ws,
                              chan_stats=self._cs_new(out) if stats else None)

    def _conv(self, pk, src16, out, ks=3, res=None, stats=True, **kw):
        if self.precision.npass == 3 and ks == 3 and out is not None and src16[0].shape[2] >= 256 and not kw:
            return self._wide3(pk, src16, out, res, stats)
        nel = out.numel() if out is not None else src16[0].numel() // src16[0].shape[-1] * kw["out16"][0].shape[-1]
        return ops.conv_igemm(None, pk.hi, pk.lo, out, prec=self.precision, ks=ks, src16=src16, bias=pk.bias, res=res, w_frag=pk.frag,
                              chan_stats=self._cs_new(out) if (stats and out is not None) else None, ws=self._ws(nel),
                              w_frag16=pk.frag16 if ks == 3 else None, **kw)

    def _res(self, tag, rb: ResnetBlock, x):
        """ResnetBlock.forward model.py:117-140 with temb None."""
        B, H, W, _ = x.shape
        co = rb.out_channels
        has_skip = rb.in_channels != rb.out_channels
        if has_skip:
            a16, x16 = self._norm16(rb.norm1, 1, x, want_raw=True)
        else:
            a16 = self._norm16(rb.norm1, 1, x)
        h = self._buf(f"h.{B}x{H}x{W}x{co}", (B, H, W, co))
        self._conv(self._packed[id(rb.conv1)], a16, h)
        out = self._buf(tag + ".out", (B, H, W, co))
        pk2 = self._packed[id(rb.conv2)]
        h16 = self._norm16(rb.norm2, 1, h, kind="h16")
        if not has_skip:
            return self._conv(pk2, h16, out, res=x)
        ps = self._packed[id(rb.nin_shortcut)]
        fkey = ("fuse", id(rb), B, H, W)
        fused = self._packed.get(fkey)
        kw = dict(prec=self.precision, src16=h16, bias=pk2.bias, w_frag=pk2.frag, chan_stats=self._cs_new(out), ws=self._ws(out.numel()),
                  w_frag16=pk2.frag16)
        if fused is None:
            fused = bool((pk2.frag is not None or pk2.frag16 is not None) and ps.frag is not None and
                         ops.conv_igemm(None, pk2.hi, pk2.lo, out, skip=(x16[0], ps.frag, ps.bias, ps.frag16), query_fused=True, **kw))
            self._packed[fkey] = fused
        if fused:
            return ops.conv_igemm(None, pk2.hi, pk2.lo, out, skip=(x16[0], ps.frag, ps.bias, ps.frag16), **kw)
        M = B * H * W        # the unfused 1x1 as one long "image" (per-pixel work: no sample structure needed, any H*W admitted)
        flat = lambda t: None if t is None else t.view(1, 1, M, t.shape[-1])
        self._conv(ps, (flat(x16[0]), flat(x16[1])), flat(out), ks=1, stats=False)
        if self.precision.npass == 3 and W >= 256:
            return self._wide3(pk2, h16, out, out, True)
        return ops.conv_igemm(None, pk2.hi, pk2.lo, out, res=out, **kw)

    def _attn(self, tag, ab: AttnBlock, x):
        """AttnBlock.forward model.py:168-199: single head of width C over T = H*W tokens, logits scaled by C^-0.5, fp32 softmax.
        Per sample: S = Q K^T and O = P V as GEMMs on the convolution kernels (K and V^T packed as their weight operand), the
        T x T logits materialised once per sample (as the reference's bmm does)."""
        prec = self.precision
        B, H, W, C = x.shape
        T = H * W
        if T % 64 != 0 or C % 64 != 0:
            raise StedmHipError(f"VQ AttnBlock: H*W = {T} and C = {C} must be multiples of 64")
        n16 = self._norm16(ab.norm, 0, x)
        q16 = self._planes((B, H, W, C), "q16")
        self._conv(self._packed[id(ab.q)], n16, None, ks=1, stats=False, out16=q16)
        k = self._buf(f"{tag}.k", (B, H, W, C)); v = self._buf(f"{tag}.v", (B, H, W, C))
        self._conv(self._packed[id(ab.k)], n16, k, ks=1, stats=False)
        self._conv(self._packed[id(ab.v)], n16, v, ks=1, stats=False)
        att = self._buf(f"{tag}.att", (B, H, W, C))
        S = self._buf("attn.S", (T, T))
        P16 = self._planes((T, T), "attn.P")
        frag_ok = prec.npass == 1

        def operand(mat, sn, sc, cout, cin):
            """(w_hi, w_lo, w_frag) of the GEMM's weight operand W[n][ci] = mat.flatten()[n * sn + ci * sc]: fragment order for the
            register-streamed kernel (its [cout][cin] planes only when a problem falls to the LDS-operand kernels), hi / lo planes in
            the 3-product mode"""
            args = (mat, sn, sc, False, cout, cin, 1, prec)
            if frag_ok:
                frag = ops.pack_conv_weight_strided(*args, want_hi=False, want_frag=True)[2]
                return ops.LazyPlanes(lambda: ops.pack_conv_weight_strided(*args, want_hi=True, want_frag=False)[:2]), None, frag
            hi, lo, _ = ops.pack_conv_weight_strided(*args, want_hi=True, want_frag=False)
            return hi, lo, None

        for b in range(B):
            kh, kl, kf = operand(k[b], C, 1, T, C)                      # W[key t][channel c] = K[t][c]
            src = (q16[0][b].view(1, 1, T, C), None if q16[1] is None else q16[1][b].view(1, 1, T, C))
            ops.conv_igemm(None, kh, kl, S.view(1, 1, T, T), prec=prec, ks=1, src16=src, w_frag=kf, ws=self._ws(T * T))
            ops.softmax_rows16(S, float(C) ** -0.5, P16[0], P16[1], prec)
            vh, vl, vf = operand(v[b], 1, C, C, T)                      # W[channel c][key t] = V[t][c]
            ops.conv_igemm(None, vh, vl, att[b].view(1, 1, T, C), prec=prec, ks=1,
                           src16=(P16[0].view(1, 1, T, T), None if P16[1] is None else P16[1].view(1, 1, T, T)), w_frag=vf, ws=self._ws(T * C))
        out = self._buf(tag + ".out", (B, H, W, C))
        return self._conv(self._packed[id(ab.proj_out)], self._norm16(None, 0, att, kind="att16"), out, ks=1, res=x)

    def _up(self, tag, up: Upsample, x):
        """Upsample.forward model.py:53-57: nearest x2 then conv 3x3, evaluated in the sub-pixel form (4 output parities x 2x2 taps)."""
        B, H, W, C = x.shape
        pk = self._packed[id(up.conv)]
        out = self._buf(tag + ".out", (B, 2 * H, 2 * W, C))
        src16 = self._norm16(None, 0, x, kind="up16")
        if W >= 256:
            # input rows of 256+ pixels (the 512^2 level of the shipped config): the sub-pixel kernel's two-plane chunk ring does not fit
            # LDS there; the nearest-x2 plane is materialised in 16 bits (a layout copy) and the plain 3x3 runs on it (9/4 of the MACs)
            up16 = self._planes((B, 2 * H, 2 * W, C), "upx2")
            for s16, d16 in zip(src16, up16):
                if s16 is not None:
                    d16.view(B, H, 2, W, 2, C).copy_(s16.view(B, H, 1, W, 1, C).expand(B, H, 2, W, 2, C))
            return self._conv(self._packed[("plain", id(up.conv))], up16, out)
        return ops.conv_igemm(None, pk.hi, pk.lo, out, prec=self.precision, mode=CONV_UP_SUBPIXEL, src16=src16,
                              bias=pk.bias, w_frag=pk.frag, chan_stats=self._cs_new(out, 4 * ops.gn_chan_nslab(H * W)))

    def _down(self, tag, dn: Downsample, x):
        """Downsample.forward model.py:69-79: zero pad bottom/right by one, conv 3x3 stride 2."""
        B, H, W, C = x.shape
        if H % 2 or W % 2:
            raise StedmHipError("VQ Downsample needs even H, W")
        pk = self._packed[id(dn.conv)]
        prec = self.precision
        one = Precision(prec.mm_dtype, 1)
        hi = self._buf(f"s2d.hi.{B}x{H}x{W}x{C}", (B, H // 2, W // 2, 4 * C), torch.int16)
        lo = self._buf(f"s2d.lo.{B}x{H}x{W}x{C}", (B, H // 2, W // 2, 4 * C), torch.int16) if prec.npass == 3 else None
        ops.space_to_depth16(x, hi, lo, prec)
        out = self._buf(tag + ".out", (B, H // 2, W // 2, C))
        ws = self._ws(out.numel())
        if prec.npass == 1:
            return ops.conv_igemm(None, None, None, out, prec=one, mode=CONV_S2D, src16=(hi, None), bias=pk.bias, w_frag=pk.frag, chan_stats=self._cs_new(out),
                                  ws=ws, pad_br=True)
        ops.conv_igemm(None, None, None, out, prec=one, mode=CONV_S2D, src16=(lo, None), bias=None, w_frag=pk.frag, ws=ws, pad_br=True)
        ops.conv_igemm(None, None, None, out, prec=one, mode=CONV_S2D, src16=(hi, None), bias=None, w_frag=pk.extra, res=out, ws=ws, pad_br=True)
        return ops.conv_igemm(None, None, None, out, prec=one, mode=CONV_S2D, src16=(hi, None), bias=pk.bias, w_frag=pk.frag, res=out,
                              chan_stats=self._cs_new(out), ws=ws, pad_br=True)

    def _maybe_attn(self, tag, attn_list, i, h):
        if len(attn_list) > 0:
            h = self._attn(f"{tag}.attn{i}", attn_list[i], h)
        return h

    # ------------------------------------------------------------------------------------------------ encoder / decoder
    @torch.no_grad()
    def _encoder(self, x):
        """Encoder.forward model.py:433-459. x [B,3,H,W] NCHW -> [B,z,H/4,W/4] NCHW."""
        enc = self.encoder
        B, _, H, W = x.shape
        h = self._buf("enc.in", (B, H, W, enc.ch))
        cs = self._cs_new(h, H // 2) if H % 2 == 0 else None
        if not ops.conv_in(x, None, enc.conv_in.weight, enc.conv_in.bias, h, chan_stats=cs) and cs is not None:
            del self._cs[h.data_ptr()]
        for i_level in range(enc.num_resolutions):
            lv = enc.down[i_level]
            for i_block in range(enc.num_res_blocks):
                h = self._res(f"enc.d{i_level}.b{i_block}", lv.block[i_block], h)
                h = self._maybe_attn(f"enc.d{i_level}", lv.attn, i_block, h)
            if i_level != enc.num_resolutions - 1:
                h = self._down(f"enc.d{i_level}.down", lv.downsample, h)
        h = self._res("enc.mid1", enc.mid.block_1, h)
        if isinstance(enc.mid.attn_1, AttnBlock):
            h = self._attn("enc.mid.attn", enc.mid.attn_1, h)
        h = self._res("enc.mid2", enc.mid.block_2, h)
        out = torch.empty((h.shape[0], enc.conv_out.out_channels, h.shape[1], h.shape[2]), dtype=torch.float32, device=h.device)
        n = enc.norm_out
        return ops.conv_out(h, n.weight, n.bias, n.eps, n.num_groups, self._out_w[id(enc.conv_out)], enc.conv_out.bias, out, self._chan_stats(h))

    @torch.no_grad()
    def _decoder(self, z):
        """Decoder.forward model.py:528-568. z [B,zc,h,w] NCHW -> [B,out_ch,4h,4w] NCHW."""
        dec = self.decoder
        B, _, H, W = z.shape
        h = self._buf("dec.in", (B, H, W, dec.conv_in.out_channels))
        # z_channels (3 or 4) -> block_in: zero-padded to 32 channels (layout shuffle) and run on the MFMA kernels like every other conv
        zp = self._buf("dec.zpad", (B, H, W, 32))
        zp.zero_()
        zp[..., :z.shape[1]].copy_(z.permute(0, 2, 3, 1))
        self._conv(self._packed[id(dec.conv_in)], self._norm16(None, 0, zp, kind="z16"), h)
        h = self._res("dec.mid1", dec.mid.block_1, h)
        if isinstance(dec.mid.attn_1, AttnBlock):
            h = self._attn("dec.mid.attn", dec.mid.attn_1, h)
        h = self._res("dec.mid2", dec.mid.block_2, h)
        for i_level in reversed(range(dec.num_resolutions)):
            lv = dec.up[i_level]
            for i_block in range(dec.num_res_blocks + 1):
                h = self._res(f"dec.u{i_level}.b{i_block}", lv.block[i_block], h)
                h = self._maybe_attn(f"dec.u{i_level}", lv.attn, i_block, h)
            if i_level != 0:
                h = self._up(f"dec.u{i_level}.up", lv.upsample, h)
        out = torch.empty((h.shape[0], dec.conv_out.out_channels, h.shape[1], h.shape[2]), dtype=torch.float32, device=h.device)
        n = dec.norm_out
        return ops.conv_out(h, n.weight, n.bias, n.eps, n.num_groups, self._out_w[id(dec.conv_out)], dec.conv_out.bias, out, self._chan_stats(h))

    # ------------------------------------------------------------------------------------------------ the reference's surface
    @torch.no_grad()
    def encode(self, x):
        """autoencoder.py:269-272: encoder + quant_conv, NO quantisation (the latents of the diffusion model are pre-quant)."""
        self._prepare()
        self._cs = {}
        h = self._encoder(x.float().contiguous())
        return ops.conv1x1_nchw(h, self.quant_conv.weight, self.quant_conv.bias)

    @torch.no_grad()
    def decode(self, h, force_not_quantize=False):
        """autoencoder.py:274-282: quantise (unless told not to), post_quant_conv, decoder."""
        self._prepare()
        self._cs = {}
        h = h.float().contiguous()
        if not force_not_quantize:
            quant, _, _ = self.quantize(h)
        else:
            quant = h
        quant = ops.conv1x1_nchw(quant, self.post_quant_conv.weight, self.post_quant_conv.bias)
        return self._decoder(quant)

    @torch.no_grad()
    def decode_code(self, code_b):
        """autoencoder.py:107-110 (through the interface's decode without re-quantising)."""
        quant_b = self.quantize.get_codebook_entry(code_b, tuple(code_b.shape) + (self.embed_dim,))
        return self.decode(quant_b, force_not_quantize=True)

    def forward(self, input, return_pred_indices=False):
        raise NotImplementedError("VQModel.forward (autoencoder training) is outside the frozen first stage")
