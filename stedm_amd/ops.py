"""Thin torch-tensor front end over the C ABI (include/stedm_hip.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every function below
forwards raw device pointers to libstedm_hip.so and raises on error. No function in this
module computes anything with torch ops."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import BF16, CONV_DOWN, CONV_S1, CONV_S2D, CONV_UP, CONV_UP_SUBPIXEL, F16, ConvArgs, check, lib


@dataclass(frozen=True)
class Precision:
    """MFMA operand format of the contraction kernels.
    parity: fp16 operands, 3 split products (hi*hi + hi*lo + lo*hi), fp32 accumulate — the mode
            checked against the oracle at 1e-3 (SURVEY.md §7 precision budget).
    fast:   single product (fp16 or bf16 operands), fp32 accumulate / activations / GN / softmax."""
    mm_dtype: int = F16
    npass: int = 3
    attn_fp8: bool = False      # style encoder's attention on e4m3 MFMA operands (BASELINE config 5); everything else as mm_dtype / npass say

    @staticmethod
    def parse(name: str) -> "Precision":
        table = {"parity": Precision(F16, 3), "parity_bf16": Precision(BF16, 3),
                 "fast": Precision(F16, 1), "f16": Precision(F16, 1), "bf16": Precision(BF16, 1),
                 "fp8": Precision(BF16, 1, True), "bf16+fp8attn": Precision(BF16, 1, True)}
        if name not in table:
            raise ValueError(f"unknown precision {name!r}; choose from {sorted(table)}")
        return table[name]

    @property
    def label(self) -> str:
        return ("f16" if self.mm_dtype == F16 else "bf16") + ("x3" if self.npass == 3 else "") + ("+fp8attn" if self.attn_fp8 else "")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, dtype=torch.float32, name="tensor"):
    if not t.is_cuda:
        raise _lib.StedmHipError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")


def device_cus() -> int:
    return lib().stedm_device_cus()


# ------------------------------------------------------------------------------------------- fp16 operand range guard
# The reference is fp32 (train_diff.py:48). The `f16` / `parity` modes round the residual stream to fp16 operand planes: a value beyond 65 504
# becomes inf there. The kernels that do the rounding flag it (include/stedm_hip.h, stedm_f16_guard_set); the host checks at its own
# synchronisation points and RAISES — a silently wrong sample is the one outcome this must exclude.
_F16_GUARD: dict = {}          # device index -> the 4 flag words (kept for the life of the process: captured graphs hold the address)
F16_GUARD_SITES = {1: "un-normalised operand planes of a 1x1 skip_connection (stedm_gn_apply16c)", 2: "GroupNorm output planes",
                   4: "16-bit side output of a convolution (Upsample operand / qkv planes)", 8: "space-to-depth planes (Downsample operand)",
                   16: "plain fp32 -> fp16 conversion (stedm_gn_apply16 / stedm_gn_chan_stats16)", 32: "fp32-source convolution loader"}


def f16_guard_enable() -> torch.Tensor:
    """Allocate (once per device) and register the guard's flag words for the current device. Not inside a graph capture."""
    dev = torch.cuda.current_device()
    w = _F16_GUARD.get(dev)
    if w is None:
        w = torch.zeros((4,), dtype=torch.int32, device=torch.device("cuda", dev))
        check(lib().stedm_f16_guard_set(w.data_ptr()), "stedm_f16_guard_set")
        _F16_GUARD[dev] = w
    return w


def f16_guard_check(where: str = "") -> None:
    """Read (synchronising) and clear the flag of the current device; raise StedmHipError when an fp16 operand overflowed since the last
    check. A no-op when no fp16-mode model ever enabled the guard on this device."""
    coop_check(where)
    w = _F16_GUARD.get(torch.cuda.current_device()) if torch.cuda.is_available() else None
    if w is None:
        return
    bits = int(w[0].item())
    if bits:
        w.zero_()
        sites = "; ".join(v for k, v in F16_GUARD_SITES.items() if bits & k)
        raise _lib.StedmHipError(
            f"fp16 operand overflow{' in ' + where if where else ''}: a value beyond the fp16 range (|x| > 65504) or a NaN was rounded into "
            f"an MFMA operand plane [{sites}]. The reference computes in fp32; run this checkpoint with precision='bf16', or 'parity_bf16' for the 1e-3 tolerance (fp32 exponent "
            f"range) — the 'f16' and 'parity' modes cannot represent its activations.")


# ------------------------------------------------------------------------------------------- in-launch GroupNorm hand-off (stedm_conv_args.gn_coop)
_COOP: dict = {}               # device index -> [int32[4] per model]: word 0 = the epoch of the model's current forward, word 1 = the give-up flag of the bounded spins


def coop_words_new() -> torch.Tensor:
    """Two device words for the cooperative GroupNorm epilogue of ONE model on the current device (a model's forwards are serial; two models
    may run on two streams). Registered for coop_check(); kept for the life of the process (captured graphs hold the addresses). Not inside a
    graph capture."""
    dev = torch.cuda.current_device()
    w = torch.zeros((4,), dtype=torch.int32, device=torch.device("cuda", dev))
    _COOP.setdefault(dev, []).append(w)
    return w


def coop_check(where: str = "") -> None:
    """Read (synchronising) and clear the give-up flags; raise when a tile stopped waiting for its sample's other tiles (their planes are wrong)."""
    ws = _COOP.get(torch.cuda.current_device(), []) if torch.cuda.is_available() else []
    bad = False
    for w in ws:
        if int(w[1].item()):
            w[1:2].zero_()
            bad = True
    if bad:
        raise _lib.StedmHipError(f"cooperative GroupNorm epilogue{' in ' + where if where else ''}: a tile gave up waiting for the channel sums of its sample's other "
                                 f"tiles (stedm_conv_args.gn_coop); the GroupNorm planes of that launch are invalid. STEDM_NO_GN_COOP=1 runs the separate pass instead.")


# ------------------------------------------------------------------------------------------- weights
def pack_conv_weight(w: torch.Tensor, prec: Precision) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """OIHW (or [O][I][1] Conv1d) fp32 -> ([O][taps][I] 16-bit hi, lo or None)."""
    if w.dim() == 3:
        w = w.unsqueeze(-1)
    w = w.detach().contiguous()
    _chk(w, name="conv weight")
    cout, cin, ks, ks2 = w.shape
    assert ks == ks2
    hi = torch.empty((cout, ks * ks, cin), dtype=torch.int16, device=w.device)
    lo = torch.empty_like(hi) if prec.npass == 3 else None
    check(lib().stedm_pack_conv_weight(w.data_ptr(), hi.data_ptr(), _ptr(lo), cout, cin, ks, prec.mm_dtype, _stream()),
          "stedm_pack_conv_weight")
    return hi, lo


def pack_conv_weight_up(w: torch.Tensor, prec: Precision) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """OIHW 3x3 fp32 -> sub-pixel upsample planes [4*O][4][I] (parity-major), see stedm_pack_conv_weight_up."""
    w = w.detach().contiguous()
    _chk(w, name="conv weight")
    cout, cin, ks, _ = w.shape
    assert ks == 3
    hi = torch.empty((4 * cout, 4, cin), dtype=torch.int16, device=w.device)
    lo = torch.empty_like(hi) if prec.npass == 3 else None
    check(lib().stedm_pack_conv_weight_up(w.data_ptr(), hi.data_ptr(), _ptr(lo), cout, cin, prec.mm_dtype, _stream()),
          "stedm_pack_conv_weight_up")
    return hi, lo


def pack_conv_weight_frag(w: torch.Tensor, prec: Precision) -> torch.Tensor:
    """OIHW 3x3 / 1x1 fp32 -> MFMA-fragment-order 16-bit weights (see stedm_pack_conv_weight_frag)."""
    w = w.detach().contiguous()
    _chk(w, name="conv weight")
    cout, cin, ks, _ = w.shape
    assert ks in (1, 3) and cin % 16 == 0
    out = torch.empty(((cout + 127) // 128, cin // 16, ks * ks, 4, 64, 8), dtype=torch.int16, device=w.device)
    check(lib().stedm_pack_conv_weight_frag(w.data_ptr(), out.data_ptr(), cout, cin, ks, prec.mm_dtype, _stream()), "stedm_pack_conv_weight_frag")
    return out


def pack_conv_weight_frag16(w: torch.Tensor, prec: Precision, sn: Optional[int] = None, sc: Optional[int] = None, flip: bool = False,
                            cout: Optional[int] = None, cin: Optional[int] = None, ks: Optional[int] = None) -> torch.Tensor:
    """3x3 / 1x1 fp32 filter -> fragment order of the 16x16x32 MFMA kind (stedm_pack_conv_weight_frag16). Default: a plain OIHW filter;
    sn / sc / flip / cout / cin / ks describe a strided source like pack_conv_weight_strided (the dgrad filter of the training step)."""
    w = w.detach()
    if sn is None:
        w = w.contiguous()
        cout, cin, ks, _ = w.shape
        assert ks in (1, 3)
        sn, sc = cin * ks * ks, ks * ks
    _chk(w, name="conv weight")
    assert cin % 32 == 0
    if prec.npass == 3:      # 3-product mode: [2] = the hi stream, then the lo stream
        out = torch.empty((2, (cout + 127) // 128, cin // 32, ks * ks, 8, 64, 8), dtype=torch.int16, device=w.device)
        fn = lib().stedm_pack_conv_weight_frag16_hl if ks == 3 else lib().stedm_pack_conv_weight_frag16_hl1
        check(fn(w.data_ptr(), sn, sc, int(flip), out.data_ptr(), cout, cin, prec.mm_dtype, _stream()), "stedm_pack_conv_weight_frag16_hl")
        return out
    out = torch.empty(((cout + 127) // 128, cin // 32, ks * ks, 8, 64, 8), dtype=torch.int16, device=w.device)
    check(lib().stedm_pack_conv_weight_frag16(w.data_ptr(), sn, sc, int(flip), out.data_ptr(), cout, cin, ks, prec.mm_dtype, _stream()),
          "stedm_pack_conv_weight_frag16")
    return out


def pack_conv_weight_up_frag(w: torch.Tensor, prec: Precision) -> torch.Tensor:
    """OIHW 3x3 fp32 -> fragment-order weights of the sub-pixel upsample form (4 parities x 4 pre-summed taps)."""
    w = w.detach().contiguous()
    _chk(w, name="conv weight")
    cout, cin, ks, _ = w.shape
    assert ks == 3 and cin % 32 == 0
    out = torch.empty((4, (cout + 127) // 128, cin // 16, 4, 4, 64, 8), dtype=torch.int16, device=w.device)
    check(lib().stedm_pack_conv_weight_up_frag(w.data_ptr(), out.data_ptr(), cout, cin, prec.mm_dtype, _stream()), "stedm_pack_conv_weight_up_frag")
    return out


def pack_conv_weight_s2d_frag(w: torch.Tensor, prec: Precision, pad_br: bool = False) -> torch.Tensor:
    """OIHW 3x3 fp32 (stride-2 Downsample.op) -> fragment-order weights of the equivalent 2x2 conv over space-to-depth planes.
    pad_br: the conv pads bottom/right only (the VQ encoder's Downsample, model.py:59-76) instead of 1 on every side."""
    w = w.detach().contiguous()
    _chk(w, name="conv weight")
    cout, cin, ks, _ = w.shape
    assert ks == 3 and cin % 8 == 0
    out = torch.empty(((cout + 127) // 128, 4 * cin // 16, 4, 4, 64, 8), dtype=torch.int16, device=w.device)
    check(lib().stedm_pack_conv_weight_s2d_frag(w.data_ptr(), out.data_ptr(), cout, cin, prec.mm_dtype, int(pad_br), _stream()), "stedm_pack_conv_weight_s2d_frag")
    return out


def pack_conv_weight_up_frag16_hl(w: torch.Tensor, prec: Precision) -> torch.Tensor:
    """The sub-pixel upsample filters as hi + lo fragment streams of the 16x16x32 kind (stedm_pack_conv_weight_up_frag16_hl): the 3-product
    modes read both, the single-product modes the hi stream (RS_SUBM, conv_rs.inc)."""
    w = w.detach().contiguous()
    _chk(w, name="conv weight")
    cout, cin, ks, _ = w.shape
    assert ks == 3 and cin % 32 == 0
    out = torch.empty((2, 4, (cout + 127) // 128, cin // 32, 4, 8, 64, 8), dtype=torch.int16, device=w.device)
    check(lib().stedm_pack_conv_weight_up_frag16_hl(w.data_ptr(), out.data_ptr(), cout, cin, prec.mm_dtype, _stream()), "stedm_pack_conv_weight_up_frag16_hl")
    return out


def pack_conv_weight_s2d_frag16_hl(w: torch.Tensor, prec: Precision, pad_br: bool = False) -> torch.Tensor:
    """The space-to-depth Downsample filter as hi + lo fragment streams of the 16x16x32 kind (3-product modes: both; single-product: hi)."""
    w = w.detach().contiguous()
    _chk(w, name="conv weight")
    cout, cin, ks, _ = w.shape
    assert ks == 3 and cin % 8 == 0
    out = torch.empty((2, (cout + 127) // 128, 4 * cin // 32, 4, 8, 64, 8), dtype=torch.int16, device=w.device)
    check(lib().stedm_pack_conv_weight_s2d_frag16_hl(w.data_ptr(), out.data_ptr(), cout, cin, prec.mm_dtype, int(pad_br), _stream()),
          "stedm_pack_conv_weight_s2d_frag16_hl")
    return out


def space_to_depth16(x: torch.Tensor, out_hi: torch.Tensor, out_lo: Optional[torch.Tensor], prec: Precision) -> None:
    """NHWC fp32 [B,H,W,C] -> 16-bit planes [B,H/2,W/2,4C] (channel block py*2+px = pixel (2y+py, 2x+px))."""
    _chk(x, name="x")
    B, H, W, C = x.shape
    assert tuple(out_hi.shape) == (B, H // 2, W // 2, 4 * C) and out_hi.dtype == torch.int16
    check(lib().stedm_space_to_depth16(x.data_ptr(), C, B, H, W, out_hi.data_ptr(), _ptr(out_lo), prec.mm_dtype, _stream()), "stedm_space_to_depth16")


def transpose(w: torch.Tensor) -> torch.Tensor:
    w = w.detach().contiguous()
    _chk(w, name="matrix")
    rows, cols = w.shape
    out = torch.empty((cols, rows), dtype=torch.float32, device=w.device)
    check(lib().stedm_transpose_f32(w.data_ptr(), out.data_ptr(), rows, cols, _stream()), "stedm_transpose_f32")
    return out


# ------------------------------------------------------------------------------------------- GroupNorm
def gn_scale_shift(x1: torch.Tensor, x2: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor, eps: float,
                   scale: torch.Tensor, shift: torch.Tensor, groups: int = 32, x2_bmod: int = 0) -> None:
    """x1 [B,H,W,c1] (+ x2 [B|bmod,H,W,c2]) NHWC -> scale/shift [B, c1+c2]."""
    _chk(x1, name="x1")
    B = x1.shape[0]
    HW = x1.numel() // (B * x1.shape[-1])
    c1 = x1.shape[-1]
    c2 = 0
    if x2 is not None:
        _chk(x2, name="x2")
        c2 = x2.shape[-1]
    assert scale.shape == (B, c1 + c2) and shift.shape == (B, c1 + c2)
    check(lib().stedm_gn_scale_shift(x1.data_ptr(), c1, _ptr(x2), c2, x2_bmod, gamma.data_ptr(), beta.data_ptr(),
                                     float(eps), groups, B, HW, scale.data_ptr(), shift.data_ptr(), _stream()),
          "stedm_gn_scale_shift")


def gn_nslab(C: int, HW: int) -> int:
    return lib().stedm_gn_nslab(C, HW)


def gn_stats(x1: torch.Tensor, x2: Optional[torch.Tensor], stats: torch.Tensor, groups: int = 32, x2_bmod: int = 0) -> None:
    """Per-slab {sum, sumsq} of the virtual concat [x1|x2] into stats (float64, >= B*gn_nslab(C,HW)*groups*2 elements)."""
    _chk(x1, name="x1")
    _chk(stats, torch.float64, "stats")
    B = x1.shape[0]
    HW = x1.numel() // (B * x1.shape[-1])
    c2 = 0 if x2 is None else x2.shape[-1]
    assert stats.numel() >= B * gn_nslab(x1.shape[-1] + c2, HW) * groups * 2
    check(lib().stedm_gn_stats(x1.data_ptr(), x1.shape[-1], _ptr(x2), c2, x2_bmod, groups, B, HW, stats.data_ptr(), _stream()),
          "stedm_gn_stats")


def gn_apply16(x1: torch.Tensor, x2: Optional[torch.Tensor], out_hi: torch.Tensor, out_lo: Optional[torch.Tensor], prec: Precision,
               gamma: Optional[torch.Tensor] = None, beta: Optional[torch.Tensor] = None, eps: float = 1e-5, groups: int = 32,
               act: int = 0, stats: Optional[torch.Tensor] = None, x2_bmod: int = 0) -> None:
    """y = act(GroupNorm([x1|x2])) (or plain conversion when gamma is None) -> 16-bit NHWC planes out_hi (/out_lo)."""
    _chk(x1, name="x1")
    B = x1.shape[0]
    HW = x1.numel() // (B * x1.shape[-1])
    c2 = 0 if x2 is None else x2.shape[-1]
    check(lib().stedm_gn_apply16(x1.data_ptr(), x1.shape[-1], _ptr(x2), c2, x2_bmod, _ptr(gamma), _ptr(beta), float(eps), groups,
                                 act, _ptr(stats), B, HW, out_hi.data_ptr(), _ptr(out_lo), prec.mm_dtype, _stream()),
          "stedm_gn_apply16")


def gn_chan_nslab(HW: int) -> int:
    return (HW + 255) // 256


def gn_chan_stats(x: torch.Tensor, out: torch.Tensor) -> None:
    """Per-(sample, slot, channel) {sum, sumsq} of an NHWC fp32 tensor -> out [B][nslab][C][2] fp32 (any nslab: the sample's
    pixels are cut into nslab runs)."""
    _chk(x, name="x")
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    assert out.dtype == torch.float32 and out.dim() == 4 and out.shape[0] == B and tuple(out.shape[2:]) == (C, 2)
    check(lib().stedm_gn_chan_stats(x.data_ptr(), C, B, HW, out.shape[1], out.data_ptr(), _stream()), "stedm_gn_chan_stats")


def gn_chan_stats16(x: torch.Tensor, out: torch.Tensor, out_hi: torch.Tensor, out_lo: Optional[torch.Tensor], prec: Precision) -> None:
    """gn_chan_stats + the plain 16-bit conversion of x (hi / lo planes of x's shape) from the same read."""
    _chk(x, name="x")
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    assert out.dtype == torch.float32 and out.dim() == 4 and out.shape[0] == B and tuple(out.shape[2:]) == (C, 2)
    assert out_hi.numel() == x.numel() and out_hi.element_size() == 2 and (out_lo is None or out_lo.numel() == x.numel())
    check(lib().stedm_gn_chan_stats16(x.data_ptr(), C, B, HW, out.shape[1], out.data_ptr(), out_hi.data_ptr(), _ptr(out_lo), prec.mm_dtype,
                                      _stream()), "stedm_gn_chan_stats16")


def gn_apply16c(x1: torch.Tensor, cs1: torch.Tensor, x2: Optional[torch.Tensor], cs2: Optional[torch.Tensor], out_hi: torch.Tensor,
                out_lo: Optional[torch.Tensor], prec: Precision, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
                groups: int = 32, act: int = 0, x2_bmod: int = 0, raw: Optional[Tuple[torch.Tensor, Optional[torch.Tensor]]] = None,
                mean_rstd: Optional[torch.Tensor] = None) -> None:
    """act(GroupNorm([x1|x2])) from channel partials -> 16-bit planes (+ optional plain conversion planes `raw`; + optional
    mean_rstd [B][groups][2] fp32, the group statistics the pass folds, kept for the training backward)."""
    _chk(x1, name="x1")
    B = x1.shape[0]
    HW = x1.numel() // (B * x1.shape[-1])
    c2 = 0 if x2 is None else x2.shape[-1]
    if mean_rstd is not None:
        assert mean_rstd.dtype == torch.float32 and mean_rstd.is_contiguous() and tuple(mean_rstd.shape) == (B, groups, 2)
    check(lib().stedm_gn_apply16c_mr(x1.data_ptr(), x1.shape[-1], cs1.data_ptr(), cs1.shape[1], _ptr(x2), c2, _ptr(cs2),
                                     0 if cs2 is None else cs2.shape[1], x2_bmod, gamma.data_ptr(),
                                     beta.data_ptr(), float(eps), groups, act, B, HW, out_hi.data_ptr(), _ptr(out_lo),
                                     None if raw is None else raw[0].data_ptr(), None if raw is None else _ptr(raw[1]),
                                     _ptr(mean_rstd), prec.mm_dtype, _stream()), "stedm_gn_apply16c_mr")


def gn_apply16c_x16(c1: int, cs1: torch.Tensor, x2: Optional[torch.Tensor], cs2: Optional[torch.Tensor], out_hi: torch.Tensor, raw_hi: torch.Tensor,
                    prec: Precision, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5, groups: int = 32, act: int = 0, x2_bmod: int = 0) -> None:
    """act(GroupNorm([x1|x2])) where x1 (c1 channels) already sits as 16-bit values in channels [0, c1) of raw_hi [B,H,W,c1+c2] (written there by
    the producing convolution: conv_igemm(out=None, out16=..., out16_stride=c1+c2)); x2 fp32 as in gn_apply16c. out_hi receives the
    normalised planes, raw_hi the plain conversion of x2. Single-product modes (stedm_gn_apply16c_x16)."""
    assert prec.npass == 1 and raw_hi.dtype == torch.int16 and out_hi.dtype == torch.int16 and raw_hi.is_contiguous() and out_hi.is_contiguous()
    B, C = raw_hi.shape[0], raw_hi.shape[-1]
    HW = raw_hi.numel() // (B * C)
    c2 = 0 if x2 is None else x2.shape[-1]
    assert C == c1 + c2 and tuple(out_hi.shape) == tuple(raw_hi.shape)
    check(lib().stedm_gn_apply16c_x16(int(c1), cs1.data_ptr(), cs1.shape[1], _ptr(x2), c2, _ptr(cs2), 0 if cs2 is None else cs2.shape[1], x2_bmod,
                                      gamma.data_ptr(), beta.data_ptr(), float(eps), groups, act, B, HW, out_hi.data_ptr(), raw_hi.data_ptr(),
                                      prec.mm_dtype, _stream()), "stedm_gn_apply16c_x16")


def gn_apply16c_x16_ok(c1: int, c2: int, groups: int = 32) -> bool:
    C = c1 + c2
    return C % 8 == 0 and c1 % 8 == 0 and c1 > 0 and 3 * C * 4 <= 48 * 1024 and 0 < groups <= 64 and C % groups == 0


# ------------------------------------------------------------------------------------------- conv
class LazyPlanes:
    """[cout][tap][cin] hi (/lo) weight planes packed on first need. The register-streamed kernel reads only the fragment-order
    weights; the planes are packed when a problem falls to the LDS-operand kernels (asked through stedm_conv_rs_ok)."""

    def __init__(self, fn):
        self._fn, self._val = fn, None

    def get(self):
        if self._val is None:
            self._val = self._fn()
        return self._val

    def reset(self):
        """the source weights changed: pack again on the next need"""
        self._val = None


class PackPlan:
    """The fragment-order weight packs of a module set, recorded as they first happen and re-run later as ONE launch
    (stedm_pack_frag_multi) into the same output tensors: after an optimizer step every convolution's packs are stale, and ~130 separate
    packs of a few microseconds each are launch-bound. Valid while the source tensors keep their storage (the caller keys on data_ptr)."""

    def __init__(self, prec: Precision):
        self.prec = prec
        self.items = []          # (w tensor (kept alive), sn, sc, flip, cout, cin, taps, m16, out)
        self._table = None
        self._blocks = 0
        self._fused = frozenset()      # indices of items the optimizer kernel refreshes itself (stedm_adamw_ema_pack)
        self._rest = None              # (table, blocks, n) of the other items
        self._fresh = None             # parameter versions at the moment the fused items were last refreshed

    def frag(self, w: torch.Tensor, sn: int, sc: int, flip: bool, cout: int, cin: int, ks: int, m16: bool) -> torch.Tensor:
        """Pack now (single-tensor launch) and remember the problem. w: fp32 tensor whose element (n, ci, tap) is flat[n*sn + ci*sc + tap']."""
        _chk(w, name="w")
        taps = ks * ks
        if m16:
            assert cin % 32 == 0
            out = torch.empty(((cout + 127) // 128, cin // 32, taps, 8, 64, 8), dtype=torch.int16, device=w.device)
            check(lib().stedm_pack_conv_weight_frag16(w.data_ptr(), sn, sc, int(flip), out.data_ptr(), cout, cin, ks, self.prec.mm_dtype, _stream()),
                  "stedm_pack_conv_weight_frag16")
        else:
            assert cin % 16 == 0
            out = torch.empty(((cout + 127) // 128, cin // 16, taps, 4, 64, 8), dtype=torch.int16, device=w.device)
            check(lib().stedm_pack_conv_weight_strided(w.data_ptr(), sn, sc, int(flip), None, None, out.data_ptr(), cout, cin, ks, self.prec.mm_dtype,
                                                       _stream()), "stedm_pack_conv_weight_strided")
        self.items.append((w, sn, sc, int(flip), cout, cin, taps, int(m16), out))
        self._table = None
        self._rest = None
        self._fresh = None
        return out

    def frag_oihw(self, w4: torch.Tensor, m16: bool) -> torch.Tensor:
        """plain OIHW filter [cout, cin, ks, ks]"""
        cout, cin, ks, _ = w4.shape
        return self.frag(w4, cin * ks * ks, ks * ks, False, cout, cin, ks, m16)

    def _build_table(self, idxs):
        import struct
        rec, blk = [], 0
        for i in idxs:
            (w, sn, sc, flip, cout, cin, taps, m16, out) = self.items[i]
            rec.append(struct.pack("<QQqqiiiiii", w.data_ptr(), out.data_ptr(), sn, sc, cout, cin, taps, flip, m16, blk))
            blk += ((cout + 127) // 128) * ((cin // 32) * 4 if m16 else (cin // 16) * 2)
        if not rec:
            return None, 0, 0
        return torch.frombuffer(bytearray(b"".join(rec)), dtype=torch.uint8).to(self.items[0][0].device), blk, len(rec)

    def set_fused(self, idxs) -> None:
        """items the optimizer kernel refreshes itself; run(versions) skips them while mark_fresh(versions) still holds"""
        self._fused = frozenset(idxs)
        self._rest = None
        self._fresh = None

    def mark_fresh(self, versions) -> None:
        self._fresh = versions

    def run(self, versions=None) -> None:
        """all recorded packs again, one launch. versions: the parameters' `_version`s now — when they are the ones mark_fresh() saw, the
        fused items already hold the current weights (written by the optimizer kernel) and only the others run"""
        if not self.items:
            return
        fresh, self._fresh = self._fresh, None
        if self._fused and versions is not None and fresh == versions:
            if self._rest is None:
                self._rest = self._build_table([i for i in range(len(self.items)) if i not in self._fused])
            table, blocks, n = self._rest
            if n:
                check(lib().stedm_pack_frag_multi(table.data_ptr(), n, blocks, self.prec.mm_dtype, _stream()), "stedm_pack_frag_multi")
            return
        if self._table is None:
            self._table, self._blocks, _ = self._build_table(range(len(self.items)))
        check(lib().stedm_pack_frag_multi(self._table.data_ptr(), len(self.items), self._blocks, self.prec.mm_dtype, _stream()), "stedm_pack_frag_multi")


def conv3x3_tiles_ok(Hout: int, Wout: int) -> bool:
    """Do the tiled 3x3 kernels have a tiling for this output grid (stedm_conv3x3_tiles_ok)? Otherwise conv_igemm takes the im2col form."""
    return bool(lib().stedm_conv3x3_tiles_ok(int(Hout), int(Wout)))


_GENERIC_COLS: dict = {}      # (device, shape) -> im2col planes of the generic-shape path (reused: one convolution at a time on a stream)


def _conv3x3_generic(src16, w_hi, w_lo, out, *, prec: Precision, mode: int, **kw):
    """A 3x3 convolution on an output grid the tiled kernels cannot tile (latent widths that are not powers of two): row-major im2col of the
    16-bit planes (stedm_im2col_rows16) + the 1x1 kind over K = 9 C with the ordinary [cout][tap][cin] weight planes. Same epilogues
    (bias / embedding / residual / statistics / the riding GroupNorm), 9x the activation bytes: a correctness path, not a fast one."""
    assert kw.get("skip") is None, "the fused skip phase belongs to the tiled kernels (stedm_conv_fused_skip_ok says so)"
    B, H, W, C = src16[0].shape
    Ho, Wo = ((H - 1) // 2 + 1, (W - 1) // 2 + 1) if mode == CONV_DOWN else ((2 * H, 2 * W) if mode == CONV_UP else (H, W))
    cols = []
    for i, pl in enumerate(src16):
        if pl is None or (i == 1 and prec.npass != 3):
            cols.append(None)
            continue
        key = (pl.device, i, B, Ho, Wo, 9 * C)
        col = _GENERIC_COLS.get(key)
        if col is None:
            if len(_GENERIC_COLS) > 16:
                _GENERIC_COLS.clear()
            col = _GENERIC_COLS[key] = torch.empty((B, Ho, Wo, 9 * C), dtype=torch.int16, device=pl.device)
        check(lib().stedm_im2col_rows16(pl.data_ptr(), col.data_ptr(), B, H, W, C, {CONV_S1: 0, CONV_DOWN: 1, CONV_UP: 2}[mode], _stream()),
              "stedm_im2col_rows16")
        cols.append(col)
    if isinstance(w_hi, LazyPlanes):
        w_hi, w_lo = w_hi.get()
    cout = w_hi.shape[0]
    assert tuple(w_hi.shape) == (cout, 9, C), (tuple(w_hi.shape), cout, C)
    for k in ("w_frag", "w_frag16", "ks", "out16_stride", "cout"):
        kw.pop(k, None)
    return conv_igemm(None, w_hi.view(cout, 1, 9 * C), None if w_lo is None else w_lo.view(cout, 1, 9 * C), out, prec=prec, ks=1, mode=CONV_S1,
                      src16=(cols[0], cols[1]), **kw)


def conv_igemm(src1: Optional[torch.Tensor], w_hi: torch.Tensor, w_lo: Optional[torch.Tensor], out: torch.Tensor, *, prec: Precision,
               ks: int = 3, mode: int = CONV_S1, src2: Optional[torch.Tensor] = None, src2_bmod: int = 0,
               scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None, act: int = 0,
               bias: Optional[torch.Tensor] = None, emb: Optional[torch.Tensor] = None, emb_offset: int = 0,
               emb_bstride: int = 0, res: Optional[torch.Tensor] = None,
               src16: Optional[Tuple[torch.Tensor, Optional[torch.Tensor]]] = None, act_out: int = 0,
               out16: Optional[Tuple[torch.Tensor, Optional[torch.Tensor]]] = None, w_frag: Optional[torch.Tensor] = None,
               chan_stats: Optional[torch.Tensor] = None,
               skip: Optional[Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]] = None, query_fused: bool = False, query_rs: bool = False,
               ws: Optional[torch.Tensor] = None, pad_br: bool = False, w_frag16: Optional[torch.Tensor] = None,
               gn_next: Optional[tuple] = None, qkv_planes: Optional[tuple] = None, ln_after: Optional[tuple] = None, coop: Optional[tuple] = None,
               out16_stride: int = 0, cout: Optional[int] = None):
    """src1 [B,Hin,Win,c1] NHWC fp32 (fused path) and/or src16 = (hi, lo) 16-bit NHWC planes [B,Hin,Win,Cin] from
    gn_apply16 (DMA path) -> out [B,Hout,Wout,cout] NHWC fp32 (see stedm_conv_igemm).
    qkv_planes = (q, k, vt, T, Tp, heads, qscale): the LSA attention's operand planes as the only output of a flat to_qkv GEMM
    (stedm_conv_args.qkv_*; out and out16 None).
    ln_after = (gamma, beta, eps, res): LayerNorm over the output row (cout <= 128) + optional fp32 residual in the epilogue
    (stedm_conv_args.ln_*): out / out16 receive LayerNorm(conv + bias) + res."""
    if src16 is not None and src1 is None and ks == 3 and mode in (CONV_S1, CONV_DOWN, CONV_UP) and not out16_stride:
        _, H_, W_, _ = src16[0].shape
        Ho_, Wo_ = ((H_ - 1) // 2 + 1, (W_ - 1) // 2 + 1) if mode == CONV_DOWN else ((2 * H_, 2 * W_) if mode == CONV_UP else (H_, W_))
        if not conv3x3_tiles_ok(Ho_, Wo_):
            if query_fused or query_rs:
                return False
            return _conv3x3_generic(src16, w_hi, w_lo, out, prec=prec, mode=mode, scale=scale, shift=shift, act=act, bias=bias, emb=emb,
                                    emb_offset=emb_offset, emb_bstride=emb_bstride, res=res, act_out=act_out, out16=out16, chan_stats=chan_stats,
                                    skip=skip, ws=ws, gn_next=gn_next)
    if out is not None:
        _chk(out, name="out")
    a = ConvArgs()
    a.act_out = act_out
    if gn_next is not None:
        # (gamma, beta, eps, groups, act, out16): the GroupNorm (+ SiLU) reading `out`, written as 16-bit planes by the same call
        g_w, g_b, g_eps, g_groups, g_act, g_out = gn_next[:6]
        g_lo = None
        if isinstance(g_out, (tuple, list)):      # (hi, lo): the 3-product modes' planes
            g_out, g_lo = g_out[0], g_out[1]
        g_mr = gn_next[6] if len(gn_next) > 6 else None
        a.gn_only = int(bool(gn_next[7])) if len(gn_next) > 7 else 0
        if g_mr is not None:
            _chk(g_mr, name="gn mean_rstd")
            assert tuple(g_mr.shape) == (out.shape[0], int(g_groups), 2)
            a.gn_mr = g_mr.data_ptr()
        _chk(g_w, name="gn gamma"); _chk(g_b, name="gn beta")
        assert g_out.dtype == torch.int16 and g_out.is_contiguous() and tuple(g_out.shape) == tuple(out.shape) and chan_stats is not None
        # the convolution's own epilogue may write these planes while other tiles still gather their input: never the planes it reads
        assert src16 is None or all(t is None or t.data_ptr() != g_out.data_ptr() for t in src16), "gn_next planes alias the convolution's input planes"
        a.gn_gamma, a.gn_beta, a.gn_eps, a.gn_groups, a.gn_act, a.gn_out16 = g_w.data_ptr(), g_b.data_ptr(), float(g_eps), int(g_groups), int(g_act), g_out.data_ptr()
        if g_lo is not None:
            assert g_lo.dtype == torch.int16 and g_lo.is_contiguous() and tuple(g_lo.shape) == tuple(out.shape)
            assert src16 is None or all(t is None or t.data_ptr() != g_lo.data_ptr() for t in src16), "gn_next lo planes alias the convolution's input planes"
            a.gn_out16_lo = g_lo.data_ptr()
        assert prec.npass == 1 or g_lo is not None, "gn_next in a 3-product mode needs the (hi, lo) planes"
        if coop is not None:       # (words, state): [B][4][128][2] int64 words of this call site (zero before first use) + the model's coop_words_new()
            cwd, cw = coop
            assert cwd.dtype == torch.int64 and cwd.is_contiguous() and tuple(cwd.shape) == (out.shape[0], 4, 128, 2)
            assert cw.dtype == torch.int32 and cw.numel() >= 2
            a.gn_coop, a.gn_coop_epoch, a.gn_coop_tmo = cwd.data_ptr(), cw.data_ptr(), cw.data_ptr() + 4
    a.pad_br = int(pad_br)
    a.w_frag16 = _ptr(w_frag16)      # npass 3: the hi + lo streams of pack_conv_weight_frag16 in that mode
    a.w_frag = _ptr(w_frag) if prec.npass == 1 else None
    a.chan_stats = _ptr(chan_stats)
    a.chan_nslab = 0 if chan_stats is None else chan_stats.shape[1]
    if ws is not None:   # fp32 workspace for the split-K form of small grids
        a.ws = ws.data_ptr()
        a.ws_floats = ws.numel()
    if skip is not None:   # fused skip_connection: (raw 16-bit planes of the block input, 1x1 weights in fragment order, bias)
        a.src16b_hi = skip[0].data_ptr()
        a.cb = skip[0].shape[-1]
        a.w_frag_b = skip[1].data_ptr()
        a.bias_b = _ptr(skip[2])
        a.w_frag_b16 = _ptr(skip[3]) if len(skip) > 3 and prec.npass == 1 else None
    if out16 is not None:
        a.out16_hi = out16[0].data_ptr()
        a.out16_lo = _ptr(out16[1]) if prec.npass == 3 else None
        if out16_stride:
            # the 16-bit output goes into channels [0, cout) of a wider plane (stedm_conv_args.out16_stride): `cout` names the convolution's width
            assert cout is not None and out is None and prec.npass == 1 and out16[0].shape[-1] == out16_stride and out16[0].is_contiguous()
            a.out16_stride = int(out16_stride)
    if ln_after is not None:
        l_g, l_b, l_eps, l_res = ln_after
        _chk(l_g, name="ln gamma"); _chk(l_b, name="ln beta")
        assert res is None and prec.npass == 1 and (out is not None or out16 is not None)
        a.ln_gamma, a.ln_beta, a.ln_eps = l_g.data_ptr(), l_b.data_ptr(), float(l_eps)
        if l_res is not None:
            _chk(l_res, name="ln residual")
            a.ln_res = l_res.data_ptr()
    if qkv_planes is not None:
        qq, qk, qv, qT, qTp, qH, qs = qkv_planes
        assert out is None and out16 is None and prec.npass == 1 and src16 is not None
        nb = src16[0].shape[2] // int(qT)
        for t_, shp in ((qq, (nb * qH, qTp, 64)), (qk, (nb * qH, qTp, 64)), (qv, (nb * qH, 64, qTp))):
            assert t_.dtype == torch.int16 and t_.is_contiguous() and tuple(t_.shape) == shp, (tuple(t_.shape), shp)
        a.qkv_q, a.qkv_k, a.qkv_vt = qq.data_ptr(), qk.data_ptr(), qv.data_ptr()
        a.qkv_T, a.qkv_Tp, a.qkv_heads, a.qkv_qscale = int(qT), int(qTp), int(qH), float(qs)
        oshape = (1, 1, src16[0].shape[2], 3 * int(qH) * 64)
    else:
        oshape = out.shape if out is not None else out16[0].shape
        if cout is not None:
            oshape = tuple(oshape[:-1]) + (int(cout),)
    if src1 is not None:
        _chk(src1, name="src1")
        B, Hin, Win, c1 = src1.shape
        a.src1 = src1.data_ptr()
        a.src2 = _ptr(src2)
        c2 = 0 if src2 is None else src2.shape[-1]
    else:
        B, Hin, Win, c1 = src16[0].shape
        c2 = 0
    if src16 is not None:
        assert src16[0].dtype == torch.int16 and src16[0].is_contiguous() and tuple(src16[0].shape) == (B, Hin, Win, c1 + c2)
        a.src16_hi = src16[0].data_ptr()
        a.src16_lo = _ptr(src16[1]) if prec.npass == 3 else None
    a.c1, a.c2, a.src2_bmod = c1, c2, src2_bmod
    a.B, a.Hin, a.Win = B, Hin, Win
    a.mode, a.ks = mode, ks
    a.scale, a.shift, a.act = _ptr(scale), _ptr(shift), act
    lazy = w_hi if isinstance(w_hi, LazyPlanes) else None
    if lazy is not None:
        w_hi, w_lo = lazy._val if lazy._val is not None else (w_frag if w_frag is not None else w_frag16, None)   # placeholder pointer until the planes are known to be needed
    a.w_hi, a.w_lo = _ptr(w_hi), (_ptr(w_lo) if prec.npass == 3 else None)
    a.bias = _ptr(bias)
    a.emb = None if emb is None else emb.data_ptr() + 4 * emb_offset
    a.emb_bstride = emb_bstride
    a.res = _ptr(res)
    a.out = _ptr(out)
    a.cout = oshape[-1]
    a.npass, a.mm_dtype = prec.npass, prec.mm_dtype
    if lazy is not None and lazy._val is None:
        if src16 is None or (w_frag is None and w_frag16 is None) or prec.npass != 1 or not lib().stedm_conv_rs_ok(C.byref(a)):
            w_hi, w_lo = lazy.get()
            a.w_hi, a.w_lo = _ptr(w_hi), (_ptr(w_lo) if prec.npass == 3 else None)
    elif mode == CONV_UP_SUBPIXEL:
        assert w_hi.shape == (4 * a.cout, 4, a.c1 + a.c2), (w_hi.shape, a.cout, a.c1, a.c2)
    elif mode == CONV_S2D:
        assert w_hi is None and (w_frag is not None or w_frag16 is not None) and src1 is None, \
            "the space-to-depth form runs on the register-streamed kernel only"
    else:
        assert w_hi.shape == (a.cout, ks * ks, a.c1 + a.c2), (w_hi.shape, a.cout, ks, a.c1, a.c2)
    if query_fused:     # capability query only: would this (fused) problem run as one kernel?
        return bool(lib().stedm_conv_fused_skip_ok(C.byref(a)))
    if query_rs:        # capability query only: would the register-streamed kernel run this problem?
        return bool(lib().stedm_conv_rs_ok(C.byref(a)))
    check(lib().stedm_conv_igemm(C.byref(a), _stream()), "stedm_conv_igemm")
    return out


def conv_in(x1: torch.Tensor, x2: Optional[torch.Tensor], w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor,
            x2_bmod: int = 0, chan_stats: Optional[torch.Tensor] = None) -> bool:
    """x1 [B,c1,H,W] NCHW (+ x2 [B,c2,H,W]) -> out [B,H,W,cout] NHWC. chan_stats [B, H/2, cout, 2] (optional): channel partials of
    `out` per pair of image rows; returns whether they were written (False: the shape took the generic path, which has no
    statistics epilogue, and the caller computes them with gn_chan_stats)."""
    _chk(x1, name="x1")
    B, c1, H, W = x1.shape
    c2 = 0 if x2 is None else x2.shape[1]
    if x2 is not None:
        _chk(x2, name="x2")
    args = (x1.data_ptr(), c1, _ptr(x2), c2, x2_bmod, w.data_ptr(), _ptr(bias), out.data_ptr(), B, H, W, out.shape[-1])
    if chan_stats is not None and H % 2 == 0:
        assert tuple(chan_stats.shape) == (B, H // 2, out.shape[-1], 2) and chan_stats.dtype == torch.float32
        rc = lib().stedm_conv_in(*args, chan_stats.data_ptr(), _stream())
        if rc == 0:
            return True
        if rc != 3:
            check(rc, "stedm_conv_in")
    check(lib().stedm_conv_in(*args, None, _stream()), "stedm_conv_in")
    return False


def conv_out_weight(w_oihw: torch.Tensor) -> torch.Tensor:
    """OIHW [cout,c,3,3] -> the layout stedm_conv_out reads: HWIO [3,3,c,cp] zero-padded to cp = 4 or 8 output channels."""
    cout, c = w_oihw.shape[:2]
    cp = 4 if cout <= 4 else 8
    w = torch.zeros((3, 3, c, cp), dtype=torch.float32, device=w_oihw.device)
    w[..., :cout] = w_oihw.detach().float().permute(2, 3, 1, 0)
    return w.contiguous()


def conv_out(src: torch.Tensor, norm_weight: torch.Tensor, norm_bias: torch.Tensor, eps: float, groups: int, w: torch.Tensor,
             bias: Optional[torch.Tensor], out: torch.Tensor, chan_stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """src [B,H,W,c] NHWC -> out [B,cout,H,W] NCHW with GroupNorm + SiLU fused into the patch load (statistics from the
    producer-side channel partials; computed here by one stedm_gn_chan_stats pass when `chan_stats` is None). `w` is the OIHW
    weight [cout,c,3,3] or, to skip the per-call permute, the tensor from conv_out_weight()."""
    _chk(src, name="src")
    B, H, W, c = src.shape
    cout = out.shape[1]
    if w.dim() == 4 and tuple(w.shape[:3]) != (3, 3, c):      # OIHW
        w = conv_out_weight(w)
    assert tuple(w.shape) == (3, 3, c, 4 if cout <= 4 else 8)
    if chan_stats is None:
        chan_stats = torch.empty((B, gn_chan_nslab(H * W), c, 2), dtype=torch.float32, device=src.device)
        gn_chan_stats(src, chan_stats)
    check(lib().stedm_conv_out(src.data_ptr(), c, chan_stats.data_ptr(), chan_stats.shape[1], norm_weight.data_ptr(), norm_bias.data_ptr(), float(eps),
                               groups, w.data_ptr(), _ptr(bias), out.data_ptr(), B, H, W, out.shape[1], _stream()), "stedm_conv_out")
    return out


# ------------------------------------------------------------------------------------------- embeddings
def time_embed(t: torch.Tensor, freqs: torch.Tensor, w0t: torch.Tensor, b0: torch.Tensor, w2t: torch.Tensor, b2: torch.Tensor,
               out: torch.Tensor, ws: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(t, torch.int64, "timesteps")
    B = t.shape[0]
    mc, ted = w0t.shape
    if ws is None:
        ws = torch.empty((B * (mc + ted),), dtype=torch.float32, device=t.device)
    assert ws.numel() >= B * (mc + ted)
    check(lib().stedm_time_embed(t.data_ptr(), freqs.data_ptr(), w0t.data_ptr(), b0.data_ptr(), w2t.data_ptr(), b2.data_ptr(),
                                 out.data_ptr(), ws.data_ptr(), B, mc, ted, _stream()), "stedm_time_embed")
    return out


def emb_proj(emb: torch.Tensor, wt: torch.Tensor, bias: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    _chk(emb, name="emb")
    B, K = emb.shape
    assert wt.shape[0] == K and out.shape == (B, wt.shape[1])
    check(lib().stedm_emb_proj(emb.data_ptr(), wt.data_ptr(), bias.data_ptr(), out.data_ptr(), B, K, wt.shape[1], _stream()),
          "stedm_emb_proj")
    return out


def linear(x: torch.Tensor, wt: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, act_in: int = 0, act_out: int = 0):
    """out = act_out(bias + act_in(x) @ wt); wt K-major [K, N]; act 0 none / 1 SiLU / 2 ReLU."""
    _chk(x, name="x")
    B, K = x.shape
    check(lib().stedm_linear(x.data_ptr(), wt.data_ptr(), _ptr(bias), out.data_ptr(), B, K, wt.shape[1], act_in, act_out, _stream()),
          "stedm_linear")
    return out


# ------------------------------------------------------------------------------------------- style path
def svit_patch_embed(img, ln_w, ln_b, eps, wt, bias, pos, cls, x, patch: int):
    _chk(img, name="style images")
    B, ns, H, W, C3 = img.shape
    assert C3 == 3
    check(lib().stedm_svit_patch_embed(img.data_ptr(), B, ns, H, W, patch, ln_w.data_ptr(), ln_b.data_ptr(), float(eps), wt.data_ptr(),
                                       bias.data_ptr(), pos.data_ptr(), cls.data_ptr(), x.data_ptr(), x.shape[-1], _stream()),
          "stedm_svit_patch_embed")
    return x


def svit_patch_ln16(img, ln_w, ln_b, eps, hi, lo, patch: int, prec: Precision):
    """patch gather + LayerNorm of SPT -> 16-bit operand planes [B*ntok, patch_dim] (the Linear then runs as a GEMM)."""
    _chk(img, name="style images")
    B, ns, H, W, C3 = img.shape
    assert C3 == 3
    check(lib().stedm_svit_patch_ln16(img.data_ptr(), B, ns, H, W, patch, ln_w.data_ptr(), ln_b.data_ptr(), float(eps), hi.data_ptr(), _ptr(lo),
                                      prec.mm_dtype, _stream()), "stedm_svit_patch_ln16")


def svit_tok_place(tok, pos, cls, x):
    """x[:, 2+t] = tok[t] + pos[2+t]; x[:, 0] = cls + pos[0]; x[:, 1] = pos[1]."""
    B, T, dim = x.shape
    check(lib().stedm_svit_tok_place(tok.data_ptr(), pos.data_ptr(), cls.data_ptr(), x.data_ptr(), B, T - 2, dim, _stream()), "stedm_svit_tok_place")


def ln_apply16(x, gamma, beta, eps, hi, lo, prec: Precision):
    _chk(x, name="x")
    rows = x.numel() // x.shape[-1]
    check(lib().stedm_ln_apply16(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps), hi.data_ptr(), _ptr(lo), rows, x.shape[-1],
                                 prec.mm_dtype, _stream()), "stedm_ln_apply16")


def qkv_pack(qkv, qscale: float, q, k, vt, B: int, T: int, Tp: int, heads: int, prec: Precision):
    """q, k, vt: (hi, lo) tuples of int16 tensors [B*heads, Tp, 64] / [B*heads, 64, Tp]."""
    check(lib().stedm_qkv_pack(qkv.data_ptr(), int(qkv.dtype == torch.int16), float(qscale), q[0].data_ptr(), _ptr(q[1]), k[0].data_ptr(), _ptr(k[1]),
                               vt[0].data_ptr(), _ptr(vt[1]), B, T, Tp, heads, prec.mm_dtype, _stream()), "stedm_qkv_pack")


def lsa_flash(q, k, vt, out, B: int, T: int, Tp: int, heads: int, prec: Precision):
    check(lib().stedm_lsa_flash(q[0].data_ptr(), _ptr(q[1]), k[0].data_ptr(), _ptr(k[1]), vt[0].data_ptr(), _ptr(vt[1]),
                                out[0].data_ptr(), _ptr(out[1]), B, T, Tp, heads, prec.npass, prec.mm_dtype, _stream()), "stedm_lsa_flash")


def lsa_flash_drop(q, k, vt, out, B: int, T: int, Tp: int, heads: int, prec: Precision, p: float, seed: int, site: int):
    """lsa_flash with train-mode dropout on the attention probabilities (vit_set.py:43, 62); mask stream: include/stedm_hip.h."""
    check(lib().stedm_lsa_flash_drop(q[0].data_ptr(), _ptr(q[1]), k[0].data_ptr(), _ptr(k[1]), vt[0].data_ptr(), _ptr(vt[1]),
                                     out[0].data_ptr(), _ptr(out[1]), B, T, Tp, heads, prec.npass, prec.mm_dtype, float(p), int(seed), int(site),
                                     _stream()), "stedm_lsa_flash_drop")


def dropout_rows(src: torch.Tensor, p: float, seed: int, site: int, prec: Precision, res: Optional[torch.Tensor] = None,
                 out: Optional[torch.Tensor] = None, hi: Optional[torch.Tensor] = None, lo: Optional[torch.Tensor] = None) -> None:
    """out = dropout(src) (+ res) as fp32 and / or 16-bit operand planes (train-mode nn.Dropout of the elementwise sViT sites)."""
    _chk(src, name="src")
    check(lib().stedm_dropout_rows(src.data_ptr(), _ptr(res), _ptr(out), _ptr(hi), _ptr(lo), src.numel(), float(p), int(seed), int(site),
                                   prec.mm_dtype, _stream()), "stedm_dropout_rows")


def qkv_pack_mx8(qkv, qscale: float, q8, qs, k8, ks, vt8, vs, B: int, T: int, Tp: int, heads: int, prec: Precision):
    """qkv (fp32 rows, or int16 rows holding the qkv GEMM's 16-bit output) -> MX-fp8 operands of lsa_flash_mx8: e4m3 bytes + one E8M0 scale per
    32 elements (include/stedm_hip.h)."""
    check(lib().stedm_qkv_pack_mx8(qkv.data_ptr(), int(qkv.dtype == torch.int16), float(qscale), q8.data_ptr(), qs.data_ptr(), k8.data_ptr(), ks.data_ptr(),
                                   vt8.data_ptr(), vs.data_ptr(), B, T, Tp, heads, prec.mm_dtype, _stream()), "stedm_qkv_pack_mx8")


def lsa_flash_mx8(q8, qs, k8, ks, vt8, vs, out16, B: int, T: int, Tp: int, heads: int, prec: Precision):
    check(lib().stedm_lsa_flash_mx8(q8.data_ptr(), qs.data_ptr(), k8.data_ptr(), ks.data_ptr(), vt8.data_ptr(), vs.data_ptr(), out16.data_ptr(), B, T, Tp,
                                    heads, prec.mm_dtype, _stream()), "stedm_lsa_flash_mx8")


def svit_head(x, pool: int, c_old, ln_w, ln_b, eps, wt, bias, out, ws: Optional[torch.Tensor] = None):
    """ws: fp32 workspace for the slab partials of the token pooling (any size >= 2 * B * dim; 1024 * dim covers every batch)."""
    B, T, dim = x.shape
    check(lib().stedm_svit_head(x.data_ptr(), B, T, dim, pool, _ptr(c_old), ln_w.data_ptr(), ln_b.data_ptr(), float(eps), wt.data_ptr(),
                                bias.data_ptr(), out.data_ptr(), out.shape[-1], _ptr(ws), 0 if ws is None else ws.numel(), _stream()), "stedm_svit_head")
    return out


def geglu16(g, hi, lo, prec: Precision):
    _chk(g, name="g")
    M = g.numel() // g.shape[-1]
    check(lib().stedm_geglu16(g.data_ptr(), hi.data_ptr(), _ptr(lo), M, g.shape[-1] // 2, prec.mm_dtype, _stream()), "stedm_geglu16")


def ln_bwd_ws_floats(rows: int, dim: int) -> int:
    return 2 * dim * lib().stedm_ln_bwd_blocks(rows)


def ln_bwd(x, dy, gamma, eps: float, dx, dgamma, dbeta, add=None, accumulate: bool = False, ws=None) -> None:
    """LayerNorm backward over the rows of x / dy [rows, dim]: dx = add + d(LN)/dx . dy; dgamma, dbeta (+)= column sums.
    ws: fp32 workspace of ln_bwd_ws_floats(rows, dim) elements (a caller's cached buffer; allocated here when absent)."""
    _chk(x, name="x"); _chk(dy, name="dy")
    dim = x.shape[-1]
    rows = x.numel() // dim
    need = ln_bwd_ws_floats(rows, dim)
    if ws is None:
        ws = torch.empty((need,), dtype=torch.float32, device=x.device)
    assert ws.dtype == torch.float32 and ws.numel() >= need and ws.is_contiguous()
    check(lib().stedm_ln_bwd(x.data_ptr(), dy.data_ptr(), gamma.data_ptr(), float(eps), _ptr(add), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                             ws.data_ptr(), rows, dim, int(accumulate), _stream()), "stedm_ln_bwd")


def geglu_bwd(g, dh, dg) -> None:
    """g [M, 2I] (value | gate), dh [M, I] -> dg [M, 2I] (GEGLU of attention.py:37-44, exact GELU)."""
    _chk(g, name="g"); _chk(dh, name="dh")
    M = g.numel() // g.shape[-1]
    check(lib().stedm_geglu_bwd(g.data_ptr(), dh.data_ptr(), dg.data_ptr(), M, g.shape[-1] // 2, _stream()), "stedm_geglu_bwd")


def agg_reduce(feats, out, n: int, mode: int):
    _chk(feats, name="features")
    Bn, Fd = feats.shape
    check(lib().stedm_agg_reduce(feats.data_ptr(), out.data_ptr(), Bn // n, n, Fd, mode, _stream()), "stedm_agg_reduce")
    return out


# ------------------------------------------------------------------------------------------- Swin-V2 embedder (non-GEMM pieces)
def swin_patch16(img: torch.Tensor, hi: torch.Tensor, lo: Optional[torch.Tensor], prec: Precision) -> None:
    """img [N, 3, H, W] fp32, ANY strides (the '(b n) c h w' view of NHWC style images is read in place) -> rows [N*H/4*W/4, 64]."""
    assert img.dtype == torch.float32 and img.is_cuda and img.dim() == 4 and img.shape[1] == 3
    N, _, H, W = img.shape
    sn, sc, sh, sw = img.stride()
    check(lib().stedm_swin_patch16(img.data_ptr(), sn, sc, sh, sw, N, H, W, hi.data_ptr(), _ptr(lo), prec.mm_dtype, _stream()), "stedm_swin_patch16")


def swin_ln(y: torch.Tensor, gamma, beta, eps: float, res: Optional[torch.Tensor], out: Optional[torch.Tensor], hi: Optional[torch.Tensor],
            lo: Optional[torch.Tensor], prec: Precision, gate: Optional[torch.Tensor] = None, rows_per_gate: int = 1) -> None:
    """out = res + LayerNorm(y) as fp32 rows and / or 16-bit operand planes (row stride hi.shape[-1] >= dim; pad columns are left alone).
    gate [rows / rows_per_gate] fp32: train-mode stochastic depth, out = res + gate[row // rows_per_gate] * LayerNorm(y)."""
    _chk(y, name="y")
    dim = y.shape[-1]
    if gate is not None:
        _chk(gate, name="gate")
        assert gate.numel() * rows_per_gate == y.numel() // dim
        check(lib().stedm_swin_ln_gated(y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps), _ptr(res), _ptr(out), _ptr(hi), _ptr(lo),
                                        y.numel() // dim, dim, dim if hi is None else hi.shape[-1], gate.data_ptr(), rows_per_gate, prec.mm_dtype,
                                        _stream()), "stedm_swin_ln_gated")
        return
    check(lib().stedm_swin_ln(y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps), _ptr(res), _ptr(out), _ptr(hi), _ptr(lo),
                              y.numel() // dim, dim, dim if hi is None else hi.shape[-1], prec.mm_dtype, _stream()), "stedm_swin_ln")


def swin_window_attn(qkv: torch.Tensor, bias_kzero: torch.Tensor, scale: torch.Tensor, rpb: torch.Tensor, hi: torch.Tensor,
                     lo: Optional[torch.Tensor], N: int, H: int, W: int, heads: int, shift: int, prec: Precision) -> None:
    """rpb [heads, 64 queries, 64 keys]; hi / lo [N*H*W, ld16 >= C]. qkv: fp32 rows, or (single-product modes) int16 rows holding the
    qkv GEMM's 16-bit output."""
    C = qkv.shape[-1] // 3
    is16 = qkv.dtype == torch.int16
    _chk(qkv, torch.int16 if is16 else torch.float32, "qkv")
    check(lib().stedm_swin_window_attn(None if is16 else qkv.data_ptr(), qkv.data_ptr() if is16 else None, bias_kzero.data_ptr(), scale.data_ptr(),
                                       rpb.data_ptr(), hi.data_ptr(), _ptr(lo), hi.shape[-1], N, H, W, C, heads, shift, prec.npass, prec.mm_dtype,
                                       _stream()), "stedm_swin_window_attn")


def swin_merge16(x: torch.Tensor, hi: torch.Tensor, lo: Optional[torch.Tensor], prec: Precision) -> None:
    _chk(x, name="x")
    N, H, W, C = x.shape
    check(lib().stedm_swin_merge16(x.data_ptr(), N, H, W, C, hi.data_ptr(), _ptr(lo), prec.mm_dtype, _stream()), "stedm_swin_merge16")


def swin_rpb(cpb: torch.Tensor, index: torch.Tensor, heads: int) -> torch.Tensor:
    """cpb [ntab, heads] fp32, index [4096] int64 (query-major) -> rpb [heads, 64 queries, 64 keys] = 16 sigmoid(cpb[index])."""
    _chk(cpb, name="cpb")
    assert index.dtype == torch.int64 and index.numel() == 4096 and index.is_cuda
    out = torch.empty((heads, 64, 64), dtype=torch.float32, device=cpb.device)
    check(lib().stedm_swin_rpb(cpb.data_ptr(), index.data_ptr(), out.data_ptr(), heads, cpb.shape[0], _stream()), "stedm_swin_rpb")
    return out


def swin_token_mean(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    _chk(x, name="x")
    N, T, C = x.shape
    check(lib().stedm_swin_token_mean(x.data_ptr(), out.data_ptr(), N, T, C, _stream()), "stedm_swin_token_mean")
    return out


def spatial_rescale(x, w, out, n_stages: int):
    _chk(x, name="x")
    B, cin, H, W = x.shape
    check(lib().stedm_spatial_rescale(x.data_ptr(), _ptr(w), out.data_ptr(), B, cin, out.shape[1], H, W, n_stages, _stream()),
          "stedm_spatial_rescale")
    return out


# ------------------------------------------------------------------------------------------- attention
def attn_legacy(qkv: torch.Tensor, out: torch.Tensor, heads: int) -> torch.Tensor:
    """qkv [B,T,heads*3*ch] -> out [B,T,heads*ch]."""
    _chk(qkv, name="qkv")
    B, T, C3 = qkv.shape
    ch = C3 // (3 * heads)
    check(lib().stedm_attn_legacy(qkv.data_ptr(), out.data_ptr(), B, T, heads, ch, _stream()), "stedm_attn_legacy")
    return out


def attn_legacy16(qkv: torch.Tensor, out16: torch.Tensor, heads: int, prec: Precision) -> torch.Tensor:
    """qkv [B,T,heads*3*ch] (fp32, or the int16 plane a conv epilogue wrote) -> out16 [B,T,heads*ch] 16-bit operand plane
    (MFMA form, single-product modes). T = 64 with ch in {32, 64, 128}; T = 64 n <= 4096 with ch in {64, 128} from the int16 plane."""
    is16 = qkv.dtype == torch.int16
    _chk(qkv, torch.int16 if is16 else torch.float32, name="qkv")
    B, T, C3 = qkv.shape
    ch = C3 // (3 * heads)
    assert out16.dtype == torch.int16 and out16.numel() == B * T * heads * ch
    check(lib().stedm_attn_legacy16(qkv.data_ptr(), 1 if is16 else 0, out16.data_ptr(), B, T, heads, ch, prec.mm_dtype, _stream()), "stedm_attn_legacy16")
    return out16


def philox_normal(rows: int, shape, seed: int, stream: int, device, sample_ids: Optional[torch.Tensor] = None, first_id: int = 0) -> torch.Tensor:
    """[rows, *shape] fp32 N(0, 1) on the device; row i depends only on (seed, stream, its sample id) (stedm_philox_normal)."""
    n = 1
    for d in shape:
        n *= int(d)
    out = torch.empty((rows,) + tuple(int(d) for d in shape), dtype=torch.float32, device=device)
    if sample_ids is not None:
        assert sample_ids.dtype == torch.int64 and sample_ids.is_cuda and sample_ids.numel() == rows and sample_ids.is_contiguous()
    check(lib().stedm_philox_normal(out.data_ptr(), int(rows), n, _ptr(sample_ids), int(first_id), int(seed) & 0xFFFFFFFFFFFFFFFF, int(stream) & 0xFFFFFFFF,
                                    _stream()), "stedm_philox_normal")
    return out


# ------------------------------------------------------------------------------------------- DDIM
def ddim_step(x: torch.Tensor, e_c: torch.Tensor, e_u: Optional[torch.Tensor], coefs: torch.Tensor, x_prev: torch.Tensor,
              pred_x0: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None,
              step_idx: Optional[torch.Tensor] = None, cfg_scale: float = 1.0, rescale_phi: float = 0.7) -> torch.Tensor:
    _chk(x, name="x")
    _chk(e_c, name="e_c")
    B, Cc, H, W = x.shape
    check(lib().stedm_ddim_step(x.data_ptr(), e_c.data_ptr(), _ptr(e_u), _ptr(noise), coefs.data_ptr(), _ptr(step_idx),
                                float(cfg_scale), float(rescale_phi), x_prev.data_ptr(), _ptr(pred_x0), B, Cc, H, W, _stream()),
          "stedm_ddim_step")
    return x_prev


def step_advance(step_idx: torch.Tensor, delta: int = 1) -> None:
    check(lib().stedm_step_advance(step_idx.data_ptr(), delta, _stream()), "stedm_step_advance")


def step_set_t(ts_table: torch.Tensor, step_idx: torch.Tensor, t_buf: torch.Tensor) -> None:
    _chk(ts_table, torch.int64, "ts_table")
    _chk(t_buf, torch.int64, "t_buf")
    check(lib().stedm_step_set_t(ts_table.data_ptr(), step_idx.data_ptr(), t_buf.data_ptr(), t_buf.shape[0], _stream()),
          "stedm_step_set_t")



# ------------------------------------------------------------------------------------------- training step (backward)
def pack_conv_weight_strided(w: torch.Tensor, sn: int, sc: int, flip: bool, cout: int, cin: int, ks: int, prec: Precision, want_hi: bool = True,
                             want_frag: bool = False):
    """(hi, lo, frag) packs of the filter whose element (n, ci, tap) is w.flatten()[n*sn + ci*sc + tap'] (see stedm_pack_conv_weight_strided)."""
    _chk(w, name="w")
    taps = ks * ks
    hi = torch.empty((cout, taps, cin), dtype=torch.int16, device=w.device) if want_hi else None
    lo = torch.empty_like(hi) if (want_hi and prec.npass == 3) else None
    frag = torch.empty(((cout + 127) // 128, cin // 16, taps, 4, 64, 8), dtype=torch.int16, device=w.device) if want_frag else None
    check(lib().stedm_pack_conv_weight_strided(w.data_ptr(), sn, sc, int(flip), _ptr(hi), _ptr(lo), _ptr(frag), cout, cin, ks, prec.mm_dtype, _stream()),
          "stedm_pack_conv_weight_strided")
    return hi, lo, frag


def gn_fold(cs1: torch.Tensor, cs2: Optional[torch.Tensor], groups: int, HW: int, eps: float, out: torch.Tensor) -> torch.Tensor:
    """chan partials of [x1|x2] -> out [B][groups][2] = {mean, rstd}."""
    B, c1 = cs1.shape[0], cs1.shape[2]
    c2 = 0 if cs2 is None else cs2.shape[2]
    assert tuple(out.shape) == (B, groups, 2)
    check(lib().stedm_gn_fold(cs1.data_ptr(), cs1.shape[1], c1, _ptr(cs2), 0 if cs2 is None else cs2.shape[1], c2, groups, B, HW, float(eps),
                              out.data_ptr(), _stream()), "stedm_gn_fold")
    return out


def gn_bwd_ws_floats(B: int, HW: int, C: int, groups: int) -> int:
    return lib().stedm_gn_bwd_ws_floats(B, HW, C, groups)


def gn_bwd(x1, x2, mean_rstd, gamma, beta, groups: int, act: int, dA, add, ws, dx1, acc1: bool, dx2, acc2: bool, dx16, prec: Precision,
           dgamma, dbeta, acc_param: bool) -> None:
    """Backward of act(GroupNorm([x1|x2])): see stedm_gn_bwd."""
    _chk(x1, name="x1"); _chk(dA, name="dA")
    B, c1 = x1.shape[0], x1.shape[-1]
    HW = x1.numel() // (B * c1)
    c2 = 0 if x2 is None else x2.shape[-1]
    assert dA.numel() == B * HW * (c1 + c2) and ws.numel() >= gn_bwd_ws_floats(B, HW, c1 + c2, groups)
    check(lib().stedm_gn_bwd(x1.data_ptr(), c1, _ptr(x2), c2, mean_rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), groups, act, dA.data_ptr(),
                             _ptr(add), B, HW, ws.data_ptr(), dx1.data_ptr(), int(acc1), _ptr(dx2), int(acc2),
                             None if dx16 is None else dx16[0].data_ptr(), None if dx16 is None else _ptr(dx16[1]), prec.mm_dtype,
                             dgamma.data_ptr(), dbeta.data_ptr(), int(acc_param), _stream()), "stedm_gn_bwd")


def im2col_t16(src16: torch.Tensor, dst16: torch.Tensor, ks: int, mode: int) -> None:
    """16-bit NHWC plane [B,Hs,Ws,C] -> [(tap*C + c)][Ppad] (dst16 [ks*ks*C, Ppad])."""
    B, Hs, Ws, Cc = src16.shape
    assert src16.dtype == torch.int16 and dst16.dtype == torch.int16 and dst16.shape[0] == ks * ks * Cc
    check(lib().stedm_im2col_t16(src16.data_ptr(), dst16.data_ptr(), B, Hs, Ws, Cc, ks, mode, dst16.shape[1], _stream()), "stedm_im2col_t16")


def wgrad_to_oihw(dw: torch.Tensor, grad: torch.Tensor, cin_ld: int, cout_ld: int, accumulate: bool, nsplit: int = 1) -> None:
    cout, cin = grad.shape[0], grad.shape[1]
    taps = grad.numel() // (cout * cin)
    _chk(grad, name="grad")
    check(lib().stedm_wgrad_to_oihw(dw.data_ptr(), grad.data_ptr(), cout, cin, taps, cin_ld, cout_ld, int(accumulate), nsplit, _stream()),
          "stedm_wgrad_to_oihw")


def wgrad3x3_plan(B: int, H: int, W: int, cin: int, cout: int) -> int:
    """split count of the direct 3x3 weight-gradient kernel for this shape, 0 when the shape is not supported"""
    ks = C.c_int(0)
    return ks.value if lib().stedm_wgrad3x3_plan(B, H, W, cin, cout, C.byref(ks)) else 0


def wgrad3x3(x16: torch.Tensor, dy16: torch.Tensor, part: torch.Tensor, prec: Precision) -> None:
    """x16 [B,H,W,cin], dy16 [B,H,W,cout] bf16 planes -> part [ksplit, 9, cin, cout] fp32 partial weight gradients"""
    B, H, W, cin = x16.shape
    cout = dy16.shape[-1]
    assert x16.dtype == torch.int16 and dy16.dtype == torch.int16 and tuple(dy16.shape[:3]) == (B, H, W) and part.dtype == torch.float32
    assert part.numel() >= wgrad3x3_plan(B, H, W, cin, cout) * 9 * cin * cout
    check(lib().stedm_wgrad3x3(x16.data_ptr(), dy16.data_ptr(), part.data_ptr(), B, H, W, cin, cout, prec.mm_dtype, _stream()), "stedm_wgrad3x3")


def wgrad3x3_oihw(x16: torch.Tensor, dy16: torch.Tensor, out: torch.Tensor, prec: Precision) -> None:
    """x16 [B,H,W,cin], dy16 [B,H,W,cout] bf16 planes -> out [ksplit, cout, cin, 3, 3] fp32: the split-K slices in the parameter's own order
    (ksplit == 1: `out` may be the gradient itself; else sum_planes adds the slices)"""
    B, H, W, cin = x16.shape
    cout = dy16.shape[-1]
    assert x16.dtype == torch.int16 and dy16.dtype == torch.int16 and tuple(dy16.shape[:3]) == (B, H, W) and out.dtype == torch.float32
    assert out.is_contiguous() and out.numel() >= wgrad3x3_plan(B, H, W, cin, cout) * 9 * cin * cout
    check(lib().stedm_wgrad3x3_oihw(x16.data_ptr(), dy16.data_ptr(), out.data_ptr(), B, H, W, cin, cout, prec.mm_dtype, _stream()), "stedm_wgrad3x3_oihw")


def sum_planes(part: torch.Tensor, out: torch.Tensor, nsplit: int, accumulate: bool = False) -> None:
    """out (+)= part[0] + ... + part[nsplit - 1] (slices of out.numel() floats each, fixed order)"""
    n = out.numel()
    assert part.dtype == torch.float32 and out.dtype == torch.float32 and out.is_contiguous() and part.numel() >= nsplit * n
    check(lib().stedm_sum_planes(part.data_ptr(), out.data_ptr(), n, int(nsplit), int(accumulate), _stream()), "stedm_sum_planes")


def wgrad1x1_plan(P: int, cin: int, cout: int) -> int:
    """split count of the direct 1x1 weight-gradient kernel for this shape, 0 when the shape is not supported"""
    ks = C.c_int(0)
    return ks.value if lib().stedm_wgrad1x1_plan(P, cin, cout, C.byref(ks)) else 0


def wgrad1x1(x16: torch.Tensor, dy16: torch.Tensor, part: torch.Tensor, prec: Precision) -> None:
    """x16 [..., cin], dy16 [..., cout] bf16 planes over the same pixels -> part [ksplit, cin, cout] fp32 partial weight gradients"""
    cin, cout = x16.shape[-1], dy16.shape[-1]
    P = x16.numel() // cin
    assert x16.dtype == torch.int16 and dy16.dtype == torch.int16 and dy16.numel() // cout == P and part.dtype == torch.float32
    assert x16.is_contiguous() and dy16.is_contiguous() and part.numel() >= wgrad1x1_plan(P, cin, cout) * cin * cout
    check(lib().stedm_wgrad1x1(x16.data_ptr(), dy16.data_ptr(), part.data_ptr(), P, cin, cout, prec.mm_dtype, _stream()), "stedm_wgrad1x1")


def chan_sum_fold(cs: torch.Tensor, per_sample: Optional[torch.Tensor], ld: int, total: Optional[torch.Tensor], accumulate: bool,
                  total2: Optional[torch.Tensor] = None) -> None:
    """channel partials -> per-sample sums and the batch total (total2: a second destination of the same total)"""
    B, nslab, Cc, _ = cs.shape
    check(lib().stedm_chan_sum_fold2(cs.data_ptr(), B, nslab, Cc, _ptr(per_sample), ld, _ptr(total), int(accumulate), _ptr(total2), _stream()),
          "stedm_chan_sum_fold2")


def sum2x2(x: torch.Tensor, out: torch.Tensor, accumulate: bool) -> None:
    B, H, W, Cc = out.shape
    assert tuple(x.shape) == (B, 2 * H, 2 * W, Cc)
    check(lib().stedm_sum2x2(x.data_ptr(), out.data_ptr(), B, H, W, Cc, int(accumulate), _stream()), "stedm_sum2x2")


def zero_insert16(x: torch.Tensor, hi: torch.Tensor, lo: Optional[torch.Tensor], prec: Precision) -> None:
    B, Ho, Wo, Cc = x.shape
    assert tuple(hi.shape) == (B, 2 * Ho, 2 * Wo, Cc)
    check(lib().stedm_zero_insert16(x.data_ptr(), hi.data_ptr(), _ptr(lo), B, Ho, Wo, Cc, prec.mm_dtype, _stream()), "stedm_zero_insert16")


def attn_legacy_bwd(qkv: torch.Tensor, d_out: torch.Tensor, d_qkv: torch.Tensor, heads: int) -> None:
    B, T, C3 = qkv.shape
    nws = lib().stedm_attn_legacy_bwd_ws_floats(B, T, heads)
    ws = torch.empty((nws,), dtype=torch.float32, device=qkv.device) if nws else None
    check(lib().stedm_attn_legacy_bwd(qkv.data_ptr(), d_out.data_ptr(), d_qkv.data_ptr(), B, T, heads, C3 // (3 * heads), _ptr(ws), _stream()),
          "stedm_attn_legacy_bwd")


def gemm_f32(A: torch.Tensor, ta: bool, Bm: torch.Tensor, tb: bool, Cm: torch.Tensor, alpha: float = 1.0, beta: float = 0.0,
             ws: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Cm = alpha * op(A) @ op(Bm) + beta * Cm on 2-D fp32 tensors (rows may be strided views: last stride must be 1)."""
    for t in (A, Bm, Cm):
        assert t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.is_cuda
    M, N = Cm.shape
    K = A.shape[0] if ta else A.shape[1]
    assert (A.shape[1] if ta else A.shape[0]) == M and (Bm.shape[0] if tb else Bm.shape[1]) == N and (Bm.shape[1] if tb else Bm.shape[0]) == K
    check(lib().stedm_gemm_f32(A.data_ptr(), A.stride(0), int(ta), Bm.data_ptr(), Bm.stride(0), int(tb), Cm.data_ptr(), Cm.stride(0), M, N, K,
                               float(alpha), float(beta), _ptr(ws), 0 if ws is None else ws.numel(), _stream()), "stedm_gemm_f32")
    return Cm


def silu(x: torch.Tensor, out: torch.Tensor, dy: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dy None: out = silu(x); else out = dy * silu'(x)."""
    _chk(x, name="x")
    check(lib().stedm_silu(x.data_ptr(), _ptr(dy), out.data_ptr(), x.numel(), 0 if dy is None else 1, _stream()), "stedm_silu")
    return out


def q_sample(x0: torch.Tensor, noise: torch.Tensor, t: torch.Tensor, sqrt_ac: torch.Tensor, sqrt_1mac: torch.Tensor) -> torch.Tensor:
    """ddpm.py:277-280: sqrt_ac[t] * x0 + sqrt_1mac[t] * noise (per-sample scalars from the fp32 schedule buffers)."""
    _chk(x0, name="x0"); _chk(noise, name="noise"); _chk(t, torch.int64, "t")
    out = torch.empty_like(x0)
    B = x0.shape[0]
    check(lib().stedm_q_sample(x0.data_ptr(), noise.data_ptr(), t.data_ptr(), sqrt_ac.data_ptr(), sqrt_1mac.data_ptr(), out.data_ptr(), B,
                               x0.numel() // B, _stream()), "stedm_q_sample")
    return out


def l1_loss(pred: torch.Tensor, target: torch.Tensor, d_pred: Optional[torch.Tensor], ws: torch.Tensor, loss: torch.Tensor, grad_scale: float = 1.0):
    _chk(pred, name="pred"); _chk(target, name="target")
    assert ws.dtype == torch.float64 and ws.numel() >= 1024 and pred.numel() == target.numel()
    check(lib().stedm_l1_loss(pred.data_ptr(), target.data_ptr(), pred.numel(), float(grad_scale), _ptr(d_pred), ws.data_ptr(), loss.data_ptr(), _stream()),
          "stedm_l1_loss")
    return loss


def spatial_rescale_wgrad(x: torch.Tensor, d_out: torch.Tensor, dw: torch.Tensor, n_stages: int, accumulate: bool = False) -> torch.Tensor:
    """channel_mapper weight gradient of the SpatialRescaler: x [B,cin,H,W], d_out [B,cout,H>>n,W>>n] -> dw [cout,cin(,1,1)]."""
    _chk(x, name="x"); _chk(d_out, name="d_out"); _chk(dw, name="dw")
    B, cin, H, W = x.shape
    cout = d_out.shape[1]
    assert dw.numel() == cout * cin and tuple(d_out.shape) == (B, cout, H >> n_stages, W >> n_stages)
    ws = torch.empty((B * cin * cout,), dtype=torch.float32, device=x.device)
    check(lib().stedm_spatial_rescale_wgrad(x.data_ptr(), d_out.data_ptr(), ws.data_ptr(), dw.data_ptr(), B, cin, cout, H, W, n_stages, int(accumulate),
                                            _stream()), "stedm_spatial_rescale_wgrad")
    return dw


def axpby(x: torch.Tensor, y: torch.Tensor, alpha: float = 1.0, beta: float = 1.0) -> torch.Tensor:
    """y = alpha * x + beta * y (flat fp32 tensors, numel % 4 == 0)."""
    _chk(x, name="x"); _chk(y, name="y")
    assert x.numel() == y.numel()
    check(lib().stedm_axpby_f32(x.data_ptr(), y.data_ptr(), x.numel(), float(alpha), float(beta), _stream()), "stedm_axpby_f32")
    return y


# Kernels that write PARAMETERS through raw pointers (no torch version bump) count here; the epoch is part of the freshness token of the
# weight packs an optimizer pass wrote itself (UNetModel.freshness_token), so any later raw-pointer writer forces a full re-pack even if it
# forgets UNetModel.invalidate().
_RAW_WRITES = [0]


def raw_write_epoch() -> int:
    return _RAW_WRITES[0]


def note_raw_write() -> None:
    _RAW_WRITES[0] += 1


def adamw_ema(table: torch.Tensor, chunk_tensor: torch.Tensor, chunk_off: torch.Tensor, lr: float, beta1: float, beta2: float, eps: float,
              weight_decay: float, step: int, ema_decay: float, grad_scale: float = 1.0) -> None:
    check(lib().stedm_adamw_ema(table.data_ptr(), chunk_tensor.data_ptr(), chunk_off.data_ptr(), chunk_tensor.numel(), float(lr), float(beta1),
                                float(beta2), float(eps), float(weight_decay), int(step), float(ema_decay), float(grad_scale), _stream()), "stedm_adamw_ema")
    note_raw_write()


def adamw_ema_pack_piece() -> Tuple[int, int]:
    """(rows, ciw): a convolution weight takes (cout / rows) * (cin / ciw) blocks of stedm_adamw_ema_pack"""
    r, c = C.c_int(0), C.c_int(0)
    check(lib().stedm_adamw_ema_pack_piece(C.byref(r), C.byref(c)), "stedm_adamw_ema_pack_piece")
    return r.value, c.value


def adamw_ema_pack(descs: torch.Tensor, ndesc: int, total_blocks: int, lr: float, beta1: float, beta2: float, eps: float, weight_decay: float,
                   step: int, ema_decay: float, grad_scale: float = 1.0) -> None:
    """AdamW + EMA over convolution weights, refreshing their fragment-order packs in the same pass (stedm_adamw_ema_pack)"""
    check(lib().stedm_adamw_ema_pack(descs.data_ptr(), int(ndesc), int(total_blocks), float(lr), float(beta1), float(beta2), float(eps),
                                     float(weight_decay), int(step), float(ema_decay), float(grad_scale), _stream()), "stedm_adamw_ema_pack")
    note_raw_write()


def adamw_ema_sched(table: torch.Tensor, chunk_tensor: torch.Tensor, chunk_off: torch.Tensor, beta1: float, beta2: float, eps: float, weight_decay: float,
                    sched: torch.Tensor, sched_idx: torch.Tensor, grad_scale: float = 1.0) -> None:
    """stedm_adamw_ema with {bias corrections, EMA decay, learning rate} read from row *sched_idx of sched [n][4] on the device (captured step)"""
    assert sched.dtype == torch.float32 and sched.is_contiguous() and sched.shape[-1] == 4 and sched_idx.dtype == torch.int32
    check(lib().stedm_adamw_ema_sched(table.data_ptr(), chunk_tensor.data_ptr(), chunk_off.data_ptr(), chunk_tensor.numel(), float(beta1), float(beta2),
                                      float(eps), float(weight_decay), sched.data_ptr(), sched_idx.data_ptr(), float(grad_scale), _stream()),
          "stedm_adamw_ema_sched")
    note_raw_write()


def adamw_ema_pack_sched(descs: torch.Tensor, ndesc: int, total_blocks: int, beta1: float, beta2: float, eps: float, weight_decay: float,
                         sched: torch.Tensor, sched_idx: torch.Tensor, grad_scale: float = 1.0) -> None:
    """stedm_adamw_ema_pack with the step-dependent scalars read on the device (see adamw_ema_sched)"""
    assert sched.dtype == torch.float32 and sched.is_contiguous() and sched.shape[-1] == 4 and sched_idx.dtype == torch.int32
    check(lib().stedm_adamw_ema_pack_sched(descs.data_ptr(), int(ndesc), int(total_blocks), float(beta1), float(beta2), float(eps), float(weight_decay),
                                           sched.data_ptr(), sched_idx.data_ptr(), float(grad_scale), _stream()), "stedm_adamw_ema_pack_sched")
    note_raw_write()


def ema_update(table: torch.Tensor, chunk_tensor: torch.Tensor, chunk_off: torch.Tensor, ema_decay: float) -> None:
    """LitEma.forward (ema.py:25-44) over the optimizer's pointer table: shadow -= (1 - decay) * (shadow - p)."""
    check(lib().stedm_ema_update(table.data_ptr(), chunk_tensor.data_ptr(), chunk_off.data_ptr(), chunk_tensor.numel(), float(ema_decay), _stream()),
          "stedm_ema_update")


# ------------------------------------------------------------------------------------------- first stage (VQ-f4)
def vq_nearest(z: torch.Tensor, codebook: torch.Tensor):
    """z [B,e,H,W] fp32, codebook [n_e,e] -> (indices int64 [B,H,W], z_q [B,e,H,W] = z + (e_idx - z)); see stedm_vq_nearest."""
    _chk(z, name="z"); _chk(codebook, name="codebook")
    B, e, H, W = z.shape
    assert codebook.shape[1] == e
    idx = torch.empty((B, H, W), dtype=torch.int64, device=z.device)
    zq = torch.empty_like(z)
    check(lib().stedm_vq_nearest(z.data_ptr(), codebook.data_ptr(), codebook.shape[0], e, B, H * W, idx.data_ptr(), zq.data_ptr(), _stream()),
          "stedm_vq_nearest")
    return idx, zq


def conv1x1_nchw(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """1x1 conv over a few channels, NCHW -> NCHW (quant_conv / post_quant_conv)."""
    _chk(x, name="x")
    B, cin, H, W = x.shape
    cout = w.shape[0]
    w2 = w.detach().float().reshape(cout, cin).contiguous()
    out = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device)
    check(lib().stedm_conv1x1_nchw(x.data_ptr(), w2.data_ptr(), _ptr(None if bias is None else bias.detach().float().contiguous()), out.data_ptr(),
                                   B, cin, cout, H * W, _stream()), "stedm_conv1x1_nchw")
    return out


def softmax_rows16(x: torch.Tensor, scale: float, hi: torch.Tensor, lo: Optional[torch.Tensor], prec: Precision) -> None:
    """x [rows, n] fp32 (row stride x.stride(0)) -> softmax(scale * x) as 16-bit planes hi (/lo) [rows, ld_out >= n], zero beyond n."""
    assert x.dim() == 2 and x.stride(1) == 1 and hi.dim() == 2 and hi.is_contiguous() and hi.shape[0] == x.shape[0] and hi.shape[1] >= x.shape[1]
    check(lib().stedm_softmax_rows16(x.data_ptr(), x.stride(0), float(scale), hi.data_ptr(), _ptr(lo), x.shape[0], x.shape[1], hi.shape[1],
                                     prec.mm_dtype, _stream()), "stedm_softmax_rows16")


# ------------------------------------------------------------------------------------------- image epilogue
def image_to_uint8(x: torch.Tensor) -> torch.Tensor:
    """((clip(x, -1, 1).permute(0, 2, 3, 1) + 1) * 127.5).astype(uint8) — x [B,C,H,W] fp32 -> [B,H,W,C] uint8 (ldm_diffusion.py:93-95)."""
    _chk(x, name="x")
    B, Cc, H, W = x.shape
    out = torch.empty((B, H, W, Cc), dtype=torch.uint8, device=x.device)
    check(lib().stedm_image_to_uint8(x.data_ptr(), out.data_ptr(), B, Cc, H, W, _stream()), "stedm_image_to_uint8")
    return out


def seg_merge(seg_nchw: torch.Tensor) -> torch.Tensor:
    """seg [B,K,H,W] one-hot -> [B,H,W,2] = {class 0, sum of the other classes} (prepare_batch, ldm_diffusion.py:52-56)."""
    _chk(seg_nchw, name="segmentation")
    B, K, H, W = seg_nchw.shape
    out = torch.empty((B, H, W, 2), dtype=torch.float32, device=seg_nchw.device)
    check(lib().stedm_seg_merge(seg_nchw.data_ptr(), out.data_ptr(), B, K, H, W, _stream()), "stedm_seg_merge")
    return out


def argmax_u8(seg: torch.Tensor) -> torch.Tensor:
    """torch.argmax(seg, dim=-1) as uint8 (ldm_diffusion.py:98): seg [..., ncls] fp32."""
    _chk(seg, name="seg")
    out = torch.empty(tuple(seg.shape[:-1]), dtype=torch.uint8, device=seg.device)
    check(lib().stedm_argmax_u8(seg.data_ptr(), out.data_ptr(), out.numel(), seg.shape[-1], _stream()), "stedm_argmax_u8")
    return out


# ------------------------------------------------------------------------------------------- graphs
class Graph:
    """hipGraph captured on the current torch stream (all buffers must be allocated beforehand)."""

    def __init__(self):
        self._exec = C.c_void_p()

    def __enter__(self):
        check(lib().stedm_graph_begin(_stream()), "stedm_graph_begin")
        return self

    def __exit__(self, et, ev, tb):
        rc = lib().stedm_graph_end(_stream(), C.byref(self._exec))
        if et is None:
            check(rc, "stedm_graph_end")
        return False

    def launch(self):
        check(lib().stedm_graph_launch(self._exec, _stream()), "stedm_graph_launch")

    def __del__(self):
        try:
            if self._exec:
                lib().stedm_graph_destroy(self._exec)
        except Exception:
            pass
