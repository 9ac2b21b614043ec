"""Training step of the U-Net on the HIP path: forward (the inference kernels, every block's intermediates kept), L1 loss,
hand-scheduled backward, AdamW + EMA.

Mirrors what autograd does for the reference in `LatentDiffusion.p_losses` / `training_step` (ddpm.py:1015-1048, 345-358) with
`torch.optim.AdamW` (modules/ldm_diffusion.py:224-234) and `LitEma` (ldm/modules/ema.py:25-44); SURVEY §8 row A15.

The backward is a reverse walk over the tape `UNetModel._forward_impl` records in training mode:
  * convolution dgrad  = the forward's MFMA convolution kernel with the flipped / transposed filter (stride-2: over the
    zero-inserted gradient; nearest-2x upsample: at the high resolution, then 2x2 sums);
  * convolution wgrad  = stride-1 3x3 (and the Upsample convs over the nearest-2x plane): the direct kernel stedm_wgrad3x3 (both
    operands from the NHWC planes, transposed in the LDS reads, all nine taps from one LDS image); other shapes: one GEMM
    dW[(tap,ci)][co] = sum_p col[(tap,ci)][p] dY^T[co][p] on the forward's kernels over transposed im2col planes (stedm_im2col_t16),
    K = B*H*W split over blocks by the register-streamed kernel's split-K, in batch chunks beyond 65 536;
  * GroupNorm+SiLU, attention, embeddings, reductions: fp32 kernels of csrc/bwd.hip;
  * 16-bit operands of the backward contractions are bf16 (fp32 exponent range: no loss scaling), single product or
    hi/lo 3-product following the forward's mode; normalised operand planes are kept by a bf16 forward, recomputed from the saved
    fp32 tensors otherwise.
Gradients are bitwise reproducible (no atomics)."""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
from types import SimpleNamespace
import torch.nn as nn

from . import ops
from ._lib import BF16, F16
from .ops import Precision
from .unet import AttentionBlock, Downsample, ResBlock, UNetModel, Upsample


def _r64(n: int) -> int:
    return (n + 63) // 64 * 64


class UNetTrainer:
    """forward / backward / optimizer step for one `UNetModel` (parameters stay the module's own `nn.Parameter`s, gradients
    land in their `.grad`, so DDP-style all-reduce and checkpointing see the usual tensors)."""

    def __init__(self, unet: UNetModel, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 ema_decay: Optional[float] = 0.9999, accumulate_grad_batches: int = 1, extra_params=()):
        self.m = unet
        # parameters outside the U-Net that the same AdamW instance updates (the reference adds cond_stage_model's when
        # cond_stage_trainable, ldm_diffusion.py:224-234); their gradients are filled by the caller; no EMA (LitEma covers `model` only)
        self.extra_params = list(extra_params)
        self.direct_wgrad = True     # False: im2col + GEMM weight gradients everywhere (tests / A/B runs set the attribute)
        self.fuse_packs = True       # the optimizer pass writes the convolution weights' fragment-order packs itself; False: separate launches (tests)
        self.accumulate_grad_batches = int(accumulate_grad_batches)      # Trainer(accumulate_grad_batches=...) of train_diff.py
        self._micro = 0
        self.lr, self.betas, self.eps, self.wd = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.ema_decay = ema_decay
        self.step_count = 0
        self.ema_updates = 0
        self._grads_ready = False
        self._dpacks: Dict[int, tuple] = {}
        self._dpacks_step: Dict[int, tuple] = {}
        self._dplan, self._dplan_key = None, None
        self._opt = None
        self._ema = None
        self._touched: Optional[list] = None
        self.bucket_mb = 256            # gradient all-reduce bucket (xGMI rings are per-link bound: few, large collectives)
        self.overlap_all_reduce = True
        # single rank, no accumulation: the optimizer of a run of parameters may start on a side stream the moment the backward has written the
        # last of their gradients (the HBM-bound AdamW / EMA / re-pack pass then runs beside the dgrad / wgrad kernels of the layers below);
        # same kernels on the same values as the pass after the backward (bitwise: tests/test_gpu_train.py). OFF by default — measured on
        # MI355X (B = 64, runs of 8 .. 512 MB): 94 % of the optimizer's kernel time ran beside backward kernels, but it doubled (1.8 -> 3.9 ms
        # per step) and the backward's kernels slowed with it: 18.96 - 19.12 ms against 18.75 ms in order. opt_bucket_mb: the run length
        self.overlap_optimizer = os.environ.get("STEDM_TRAIN_OVERLAP_OPT", "0") != "0"
        self.opt_bucket_mb = int(os.environ.get("STEDM_TRAIN_OPT_BUCKET_MB", "32"))
        self.overlap_opt_fires = 0
        self.overlap_fires = 0           # all-reduces started from inside a backward so far
        self.direct_wgrad1 = True    # False: the 1x1 convolutions' weight gradients in the GEMM form
        self.wgrad_oihw = True       # the direct 3x3 kernel writes its slices in the parameter's order (False: [tap][ci][co] partials + transposing reduce)
        self.G: Dict[int, torch.Tensor] = {}

    # ------------------------------------------------------------------------------------------------ helpers
    @property
    def bprec(self) -> Precision:
        return Precision(BF16, self.m.precision.npass)

    def _buf(self, name, shape, dtype=torch.float32):
        return self.m._buf("bw." + name, shape, dtype)

    def _planes(self, kind: str, shape) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        hi = self._buf(f"{kind}.hi.{tuple(shape)}", shape, torch.int16)
        lo = self._buf(f"{kind}.lo.{tuple(shape)}", shape, torch.int16) if self.bprec.npass == 3 else None
        return hi, lo

    def _cast16(self, x1, x2=None, kind="c16"):
        """plain conversion of [x1|x2] (NHWC fp32) to the backward's 16-bit operand planes"""
        shape = tuple(x1.shape[:-1]) + (x1.shape[-1] + (0 if x2 is None else x2.shape[-1]),)
        hi, lo = self._planes(kind, shape)
        ops.gn_apply16(x1, x2, hi, lo, self.bprec)
        return hi, lo

    def _norm16(self, norm: nn.GroupNorm, act: int, x1, x2=None):
        m = self.m
        kept = m._saved16.get((id(norm), x1.data_ptr()))
        if kept is not None and self.bprec.npass == 1:       # the forward ran in the backward's operand format and kept this plane
            return kept
        shape = tuple(x1.shape[:-1]) + (x1.shape[-1] + (0 if x2 is None else x2.shape[-1]),)
        hi, lo = self._planes("a16", shape)
        ops.gn_apply16c(x1, m._chan_stats(x1), x2, None if x2 is None else m._chan_stats(x2), hi, lo, self.bprec, norm.weight, norm.bias,
                        norm.eps, norm.num_groups, act)
        return hi, lo

    def _grad_of(self, t: torch.Tensor) -> Tuple[torch.Tensor, bool]:
        """(gradient buffer of activation `t`, whether it already holds a contribution)"""
        key = t.data_ptr()
        g = self.G.get(key)
        if g is not None:
            return g, True
        g = self._buf(f"g.{key}", tuple(t.shape))
        self.G[key] = g
        return g, False

    def _alloc_grads(self) -> None:
        """Every parameter's `.grad` is a view into one flat fp32 arena (bucketed all-reduce and the optimizer kernel walk it without
        copies); the emb_layers weights come first, in the order of the concatenated embedding Linear, so that its weight
        gradient is written by one GEMM straight into place."""
        m = self.m
        emb_w = [rb.emb_layers[1].weight for rb, _ in m._emb_layout]
        seen = {id(p) for p in emb_w}
        order = emb_w + [p for p in m.parameters() if id(p) not in seen] + self.extra_params
        self._n_unet_params = len(order) - len(self.extra_params)
        offs, off = [], 0
        for i, p in enumerate(order):
            off = (off + 3) // 4 * 4              # every gradient starts 16-byte aligned: the optimizer kernels read float4 (the emb_layers
            offs.append(off)                      # weights stay contiguous — their row blocks are multiples of 4 floats; gaps stay zero)
            off += p.numel()
        total = (off + 3) // 4 * 4
        self.grad_arena = torch.zeros((total,), dtype=torch.float32, device=order[0].device)
        self._arena_off = offs
        for p, o in zip(order, offs):
            p.grad = self.grad_arena[o:o + p.numel()].view(p.shape)
        ted = m.model_channels * 4
        assert not emb_w or offs[len(emb_w) - 1] + emb_w[-1].numel() == m._emb_ntot * ted, "emb_layers weight gradients are not one contiguous matrix"
        self._dWcat = self.grad_arena[:m._emb_ntot * ted].view(m._emb_ntot, ted)
        self._arena_params = order
        self._int_views = [p.grad for p in order]
        self._pub_arena = None
        from .parallel import BucketSchedule
        self._pidx = {id(p): i for i, p in enumerate(order)}
        self._sched = BucketSchedule(self._arena_off, [p.numel() for p in order], self.bucket_mb * (1 << 20) // 4, tail_from=self._n_unet_params,
                                     align=4, total=total)

    def _param_grad(self, p: nn.Parameter) -> torch.Tensor:
        if getattr(self, "grad_arena", None) is None:
            self._alloc_grads()
        elif p.grad is None:          # e.g. after optimizer.zero_grad(set_to_none=True): the arena stays, the views come back
            self.internal_grads()
        if self._touched is not None:
            self._touched.append(p)
        return p.grad

    def _w4(self, conv) -> torch.Tensor:
        w = conv.weight.detach().float()
        if w.dim() == 2:          # nn.Linear (SpatialTransformer): a 1x1 convolution over the token rows
            return w[:, :, None, None]
        return w.unsqueeze(-1) if w.dim() == 3 else w

    def _dpack(self, conv, pad_cin: int = 0, pad_cout: int = 0):
        """weights of the dgrad convolution: filter flipped and transposed (layout shuffle), packed like forward weights"""
        key = id(conv)
        hit = self._dpacks.get(key)
        if hit is None:
            hit = self._dpacks_step.get(key)
        if hit is not None:
            return hit
        bp = self.bprec
        w = self._w4(conv).contiguous()
        # the plan re-reads the parameter's own storage on later steps: only a module's parameter qualifies. A throw-away holder (the
        # stacked q | k | v filter of _st_bwd) is its own storage too, but lives for one backward: it is packed for this step only
        in_place = isinstance(conv, nn.Module) and w.data_ptr() == conv.weight.data_ptr()
        co_f, ci_f, ks = w.shape[0], w.shape[1], w.shape[-1]
        taps = ks * ks
        frag16 = None
        if pad_cin or pad_cout:    # first / last convolution (7 and 4 channels): padded copy (layout shuffle), generic kernels
            wt = w.flip(2, 3).transpose(0, 1).contiguous()
            full = torch.zeros((pad_cout or wt.shape[0], pad_cin or wt.shape[1]) + tuple(wt.shape[2:]), dtype=torch.float32, device=wt.device)
            full[:wt.shape[0], :wt.shape[1]] = wt
            hi, lo = ops.pack_conv_weight(full, bp)
            frag = None
        else:
            # packed straight from the OIHW parameter: row n = forward input channel, column ci = forward output channel, taps reversed
            want_frag = self.m.conv_path == "dma" and bp.npass == 1 and ((ks == 3 and co_f % 16 == 0) or (ks == 1 and co_f % 64 == 0))
            args = (w, taps, ci_f * taps, True, ci_f, co_f, ks, bp)
            if want_frag:   # the planes are only read when a problem falls to the LDS-operand kernels: packed on first need
                # the dgrad conv contracts over the forward's OUTPUT channels: the 16x16x32 MFMA kind takes it from 256 of them on
                m16 = ks == 3 and self.m._m16 and co_f % 32 == 0 and co_f >= 256
                if m16:
                    frag = None
                    frag16 = (self._dplan.frag(w, taps, ci_f * taps, True, ci_f, co_f, 3, True) if in_place else
                              ops.pack_conv_weight_frag16(w, bp, sn=taps, sc=ci_f * taps, flip=True, cout=ci_f, cin=co_f, ks=3))
                else:
                    frag = (self._dplan.frag(w, taps, ci_f * taps, True, ci_f, co_f, ks, False) if in_place else
                            ops.pack_conv_weight_strided(*args, want_hi=False, want_frag=True)[2])
                hi, lo = ops.LazyPlanes(lambda: ops.pack_conv_weight_strided(*args, want_hi=True, want_frag=False)[:2]), None
                if in_place:      # kept across steps: the plan refreshes the fragments in one launch, the lazy planes are reset
                    self._dpacks[key] = (hi, lo, frag, ks, frag16)
                    return self._dpacks[key]
            else:
                hi, lo, frag = ops.pack_conv_weight_strided(*args, want_hi=True, want_frag=False)
        self._dpacks_step[key] = (hi, lo, frag, ks, frag16)
        return self._dpacks_step[key]

    def _cus(self) -> int:
        if not hasattr(self, "_ncus"):
            self._ncus = ops.device_cus() or 256
        return self._ncus

    def _ws(self, nel: int) -> Optional[torch.Tensor]:
        if nel > (1 << 23):
            return None
        return self.m._buf("conv_ws", ((16 if nel <= (1 << 20) else (4 if nel <= (1 << 22) else 2)) * nel,))

    def _dgrad(self, conv, dy16, out: torch.Tensor, accumulate: bool = False, pad_cin: int = 0, pad_cout: int = 0) -> torch.Tensor:
        """out (+)= conv(dy16, flipped filter); dy16 planes are at the resolution of `out`"""
        hi, lo, frag, ks, frag16 = self._dpack(conv, pad_cin, pad_cout)
        ops.conv_igemm(None, hi, lo, out, prec=self.bprec, ks=ks, src16=dy16, w_frag=frag, w_frag16=frag16, res=out if accumulate else None,
                       ws=self._ws(out.numel()))
        return out

    def _wgrad(self, src16, dy16, dy_f32: Optional[torch.Tensor], wparam: Optional[nn.Parameter], ks: int, mode: int,
               grad: Optional[torch.Tensor] = None) -> None:
        """wparam.grad = sum_p src[p + tap] (x) dy[p] (see module docstring); src16 / dy16 are (hi, lo) NHWC planes. grad: write there instead
        (a filter that is not one parameter: the stacked q | k | v rows of the SpatialTransformer's attentions)"""
        bp = self.bprec
        gdst = (lambda: grad) if grad is not None else (lambda: self._param_grad(wparam))
        B, Hs, Ws, Cs = src16[0].shape
        Bo, Ho, Wo, co = dy16[0].shape
        if ks == 3 and mode == 0 and bp.npass == 1 and self.m.conv_path == "dma" and self.direct_wgrad:
            nsplit = ops.wgrad3x3_plan(B, Hs, Ws, Cs, co)
            if nsplit > 0:    # direct kernel: both operands straight from the NHWC planes, transposed in the LDS reads
                g = gdst()
                if self.wgrad_oihw and g.is_contiguous() and g.data_ptr() % 16 == 0:
                    # the kernel writes in the parameter's own order: one slice lands in the gradient itself, several are summed by a
                    # streaming pass (no transposing reduce over [split][tap][ci][co] partials)
                    if nsplit == 1:
                        ops.wgrad3x3_oihw(src16[0], dy16[0], g, bp)
                    else:
                        part = self._buf("wg.part", (nsplit * 9 * Cs * co,))
                        ops.wgrad3x3_oihw(src16[0], dy16[0], part, bp)
                        ops.sum_planes(part, g, nsplit)
                    return
                part = self._buf("wg.part", (nsplit * 9 * Cs * co,))
                ops.wgrad3x3(src16[0], dy16[0], part, bp)
                ops.wgrad_to_oihw(part, g, Cs, co, False, nsplit)
                return
        if ks == 1 and mode == 0 and bp.npass == 1 and self.m.conv_path == "dma" and self.direct_wgrad and self.direct_wgrad1 and src16[0].is_contiguous() and dy16[0].is_contiguous():
            nsplit = ops.wgrad1x1_plan(B * Hs * Ws, Cs, co)
            if nsplit > 0:    # direct kernel for the 1x1 convolutions: flat [P][C] planes, no transposes through HBM
                part = self._buf("wg.part1", (nsplit * Cs * co,))
                ops.wgrad1x1(src16[0], dy16[0], part, bp)
                ops.wgrad_to_oihw(part, gdst(), Cs, co, False, nsplit)
                return
        # GEMM form. Its contraction length K = (samples) * Ho * Wo is bounded by the conv kernels' zero page (65 536): larger problems go in
        # batch chunks that accumulate into the gradient
        Bc = max(1, 65536 // (Ho * Wo))
        for b0 in range(0, Bo, Bc):
            b1 = min(Bo, b0 + Bc)
            self._wgrad_gemm(tuple(None if t is None else t[b0:b1] for t in src16), tuple(None if t is None else t[b0:b1] for t in dy16),
                             None if dy_f32 is None else dy_f32[b0:b1], wparam, ks, mode, accumulate=b0 > 0, grad=grad)

    def _wgrad_gemm(self, src16, dy16, dy_f32, wparam, ks: int, mode: int, accumulate: bool, grad: Optional[torch.Tensor] = None) -> None:
        bp = self.bprec
        B, Hs, Ws, Cs = src16[0].shape
        Bo, Ho, Wo, co = dy16[0].shape
        P = Bo * Ho * Wo
        Ppad = _r64(P)
        taps = ks * ks
        col = self._planes("col", (taps * Cs, Ppad))
        dyt = self._planes("dyt", (co, Ppad))
        for i in range(2 if bp.npass == 3 else 1):
            ops.im2col_t16(src16[i], col[i], ks, mode)
            ops.im2col_t16(dy16[i], dyt[i], 1, 0)
        frag = None
        if self.m.conv_path == "dma" and bp.npass == 1 and P == Ppad and dy_f32 is not None and dy_f32.numel() == P * co:
            # register-streamed kernel (splits K = B*H*W over blocks): dY^T in MFMA-fragment order
            frag = ops.pack_conv_weight_strided(dy_f32, 1, co, False, co, P, 1, bp, want_hi=False, want_frag=True)[2]
        dw = self._buf("dw", (taps, Cs, 1, co))
        rows = taps * Cs
        w_hi = dyt[0].view(co, 1, Ppad)
        w_lo = None if dyt[1] is None else dyt[1].view(co, 1, Ppad)

        def gemm(r0, r1):
            n = r1 - r0
            src = (col[0][r0:r1].view(1, n, 1, Ppad), None if col[1] is None else col[1][r0:r1].view(1, n, 1, Ppad))
            ops.conv_igemm(None, w_hi, w_lo, dw.view(rows, co)[r0:r1].view(1, n, 1, co), prec=bp, ks=1, src16=src, w_frag=frag, ws=self._ws(dw.numel()))

        # tile-count-aware launch split: the kernel runs one 256 x 128 tile per workgroup, one workgroup per CU; a grid of, say, 288 tiles
        # costs two full rounds. Rows that fill whole rounds go first, the remainder goes separately (where split-K refills the chip).
        tm, tn, cus = (rows + 255) // 256, (co + 127) // 128, self._cus()
        total = tm * tn
        m_full = (total // cus) * cus // tn
        if frag is not None and total > cus and 0 < total % cus < 0.6 * cus and 0 < m_full < tm:
            gemm(0, m_full * 256)
            gemm(m_full * 256, rows)
        else:
            gemm(0, rows)
        ops.wgrad_to_oihw(dw, grad if grad is not None else self._param_grad(wparam), Cs, co, accumulate)

    def _bias_grad(self, dy: torch.Tensor, bias: Optional[nn.Parameter], per_sample: Optional[torch.Tensor] = None, ld: int = 0,
                   cast: Optional[str] = None, also: Optional[nn.Parameter] = None):
        """bias.grad = sum over (batch, pixels) of dy [B,H,W,C]; per_sample[b*ld + c] = sum over pixels (optional). `cast`: the same read
        of dy also writes its 16-bit operand planes (buffer family `cast`, as _cast16 would) and returns them."""
        B, Cc = dy.shape[0], dy.shape[-1]
        cs = self._buf(f"cs.{B}x{Cc}x{dy.numel() // (B * Cc)}", (B, ops.gn_chan_nslab(dy.numel() // (B * Cc)), Cc, 2))
        planes = None
        if cast is not None:
            planes = self._planes(cast, tuple(dy.shape))
            ops.gn_chan_stats16(dy, cs, planes[0], planes[1], self.bprec)
        else:
            ops.gn_chan_stats(dy, cs)
        # `also`: a second bias added onto the same tensor receives the same sums (written by the same launch, no copy)
        ops.chan_sum_fold(cs, per_sample, ld, None if bias is None else self._param_grad(bias), False,
                          None if also is None else self._param_grad(also))
        return planes

    def _gn_bwd(self, norm: nn.GroupNorm, act: int, x1, x2, dA, add, dx16=None):
        m = self.m
        B, c1 = x1.shape[0], x1.shape[-1]
        c2 = 0 if x2 is None else x2.shape[-1]
        HW = x1.numel() // (B * c1)
        G = norm.num_groups
        mr = getattr(m, "_saved_mr", {}).get((id(norm), x1.data_ptr()))   # left by the forward's normalisation pass
        if mr is None:
            mr = self._buf(f"mr.{B}x{G}", (B, G, 2))
            ops.gn_fold(m._chan_stats(x1), None if x2 is None else m._chan_stats(x2), G, HW, norm.eps, mr)
        ws = self._buf("gnws", (ops.gn_bwd_ws_floats(B, HW, c1 + c2, G),))
        g1, a1 = self._grad_of(x1)
        g2, a2 = self._grad_of(x2) if x2 is not None else (None, False)
        ops.gn_bwd(x1, x2, mr, norm.weight, norm.bias, G, act, dA, add, ws, g1, a1, g2, a2, dx16, self.bprec, self._param_grad(norm.weight),
                   self._param_grad(norm.bias), False)

    # ------------------------------------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, x: torch.Tensor, c_concat: Optional[torch.Tensor], t: torch.Tensor, context: torch.Tensor) -> torch.Tensor:
        """eps prediction [B,out,H,W] (NCHW), keeping what the backward needs."""
        m = self.m
        m._tape = []
        try:
            out = m._forward_impl(x, c_concat, t, [context], None, uniform_t=False)
        except Exception:
            m._tape = None
            raise
        self.tape = m._tape
        self.tape_emb = m._tape_emb
        m._tape = None
        if self.ema_decay is not None and getattr(self, "_ema", None) is None:
            self._build_ema()        # LitEma clones the parameters when it is built (ema.py:17-21): before anything updates them
        return out

    # ------------------------------------------------------------------------------------------------ backward
    @torch.no_grad()
    def backward(self, d_eps: torch.Tensor, on_bucket=None, sched=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """d_eps [B,out,H,W] = dL/d(eps prediction). Fills `.grad` of every parameter; returns (dL/dx [B,in,H,W] over the
        concatenated [x | c_concat] input, dL/dcontext [B, 4*model_channels]). on_bucket(b): called as soon as every gradient of
        bucket b of the arena (self._sched.bounds[b]) is final — the walk runs from the output to the input, so the arena's buckets
        complete back to front and their all-reduce overlaps the rest of the backward."""
        m = self.m
        self.G = {}
        self._st_holders = []
        # dgrad weight packs: the fragment-order ones recorded by the last backward run again as ONE launch when the parameters kept their
        # storage (ops.PackPlan); everything else is packed when first needed, as before
        dkey = (self.bprec, m.conv_path, m._m16, tuple(p.data_ptr() for p in m.parameters()))
        self._dpacks_step = {}
        if getattr(self, "_dplan", None) is not None and self._dplan_key == dkey:
            self._dplan.run(versions=m.freshness_token())
            for ent in self._dpacks.values():
                if isinstance(ent[0], ops.LazyPlanes):
                    ent[0].reset()
        else:
            self._dpacks = {}
            self._dplan, self._dplan_key = ops.PackPlan(self.bprec), dkey
        self.internal_grads()
        sched = self._sched if sched is None else sched       # (the all-reduce buckets, or the overlapped optimizer's shorter runs)
        if on_bucket is not None:
            sched.reset()
        B = d_eps.shape[0]
        ted = m.model_channels * 4
        self.dE = self._buf("dE", (B, m._emb_ntot))
        self.dEs = None
        dx_in = None
        def flush():
            if on_bucket is not None:
                for p in self._touched:
                    b = sched.done(self._pidx[id(p)])
                    if b is not None:
                        on_bucket(b)
                self._touched = []

        self._touched = [] if on_bucket is not None else None
        for rec in reversed(self.tape):
            flush()           # a record's gradients are final once the NEXT record starts (emb_layers biases are copies made inside the record)
            kind = rec[0]
            if kind == "conv_out":
                self._conv_out_bwd(rec[1], d_eps.float().contiguous())
            elif kind == "res":
                self._res_bwd(*rec[1:])
            elif kind == "attn":
                self._attn_bwd(*rec[1:])
            elif kind == "st":
                self._st_bwd(*rec[1:])
            elif kind == "up":
                self._up_bwd(*rec[1:])
            elif kind == "down":
                self._down_bwd(*rec[1:])
            elif kind == "conv_in":
                dx_in = self._conv_in_bwd(*rec[1:])
            else:
                raise RuntimeError(kind)
        dctx = self._emb_bwd(B, ted)
        flush()
        self._touched = None
        self._grads_ready = True
        return dx_in, dctx

    def _conv_out_bwd(self, h, d_eps):
        m = self.m
        gn, conv = m.out[0], m.out[2]
        B, H, W, Cc = h.shape
        co = conv.out_channels
        cp = 32
        dy = self._buf("out.dy", (B, H, W, cp))
        dy.zero_()
        dy[..., :co].copy_(d_eps.permute(0, 2, 3, 1))            # layout shuffle NCHW -> NHWC (padded to 32 channels)
        dy16 = self._cast16(dy, kind="dy")
        a16 = self._norm16(gn, 1, h)
        self._wgrad(a16, dy16, dy, conv.weight, 3, 0)
        tot = self._buf("out.db", (cp,))
        cs = self._buf("out.cs", (B, ops.gn_chan_nslab(H * W), cp, 2))
        ops.gn_chan_stats(dy, cs)
        ops.chan_sum_fold(cs, None, 0, tot, False)
        self._param_grad(conv.bias).copy_(tot[:co])
        dA = self._buf(f"dA.{B}x{H}x{W}x{Cc}", (B, H, W, Cc))
        self._dgrad(conv, dy16, dA, pad_cin=cp)
        self._gn_bwd(gn, 1, h, None, dA, None)

    def _conv_in_bwd(self, x, c_concat, h0):
        m = self.m
        conv = m.input_blocks[0][0]
        B, H, W, co = h0.shape
        cin = conv.in_channels
        cp = 32
        g, have = self._grad_of(h0)
        assert have
        dy16 = self._cast16(g, kind="dy")
        xin = self._buf("in.x", (B, H, W, cp))
        xin.zero_()
        xin[..., :x.shape[1]].copy_(x.permute(0, 2, 3, 1))
        if c_concat is not None:
            xin[..., x.shape[1]:cin].copy_(c_concat.permute(0, 2, 3, 1))
        x16 = self._cast16(xin, kind="x16")
        self._wgrad(x16, dy16, g, conv.weight, 3, 0)
        self._bias_grad(g, conv.bias)
        dxp = self._buf("in.dx", (B, H, W, cp))
        self._dgrad(conv, dy16, dxp, pad_cout=cp)
        return dxp[..., :cin].permute(0, 3, 1, 2).contiguous()

    def _res_bwd(self, rb: ResBlock, x1, x2, h, out, emb_off):
        B, H, W, co = out.shape
        cin = x1.shape[-1] + (0 if x2 is None else x2.shape[-1])
        conv1, conv2 = rb.in_layers[2], rb.out_layers[3]
        dout, have = self._grad_of(out)
        assert have, "no gradient reached this block's output"
        has_skip = not isinstance(rb.skip_connection, nn.Identity)
        dout16 = self._bias_grad(dout, conv2.bias, cast="dy", also=rb.skip_connection.bias if has_skip else None)   # both biases add onto `out`
        if has_skip:
            sk = rb.skip_connection
            x16 = self.m._saved16.get(("raw", x1.data_ptr())) or self._cast16(x1, x2, kind="x16")
            self._wgrad(x16, dout16, dout, sk.weight, 1, 0)
            add = self._buf(f"add.{B}x{H}x{W}x{cin}", (B, H, W, cin))
            self._dgrad(sk, dout16, add)
        else:
            add = dout
        # conv2 and its GroupNorm + SiLU
        h16 = self._norm16(rb.out_layers[0], 1, h)
        self._wgrad(h16, dout16, dout, conv2.weight, 3, 0)
        dA2 = self._buf(f"dA.{B}x{H}x{W}x{co}", (B, H, W, co))
        self._dgrad(conv2, dout16, dA2)
        dh16 = self._planes("dh16", (B, H, W, co))
        self._gn_bwd(rb.out_layers[0], 1, h, None, dA2, None, dx16=dh16)
        dh = self.G[h.data_ptr()]
        # the embedding enters between conv1 and the second GroupNorm (openaimodel.py:276-287): its gradient is the per-sample
        # channel sum of dh, conv1's bias gradient the batch total
        eb = rb.emb_layers[1].bias                                        # same sums: both biases add onto h
        if emb_off is None:
            self.dEs = self._buf("dEs", (B, co))
            self._bias_grad(dh, conv1.bias, self.dEs, co, also=eb)
        else:
            self._bias_grad(dh, conv1.bias, self.dE[:, emb_off:], self.dE.shape[1], also=eb)
        # conv1 and the first GroupNorm + SiLU over the (virtual concat) input
        a16 = self._norm16(rb.in_layers[0], 1, x1, x2)
        self._wgrad(a16, dh16, dh, conv1.weight, 3, 0)
        dA1 = self._buf(f"dA1.{B}x{H}x{W}x{cin}", (B, H, W, cin))
        self._dgrad(conv1, dh16, dA1)
        self._gn_bwd(rb.in_layers[0], 1, x1, x2, dA1, add)

    def _attn_bwd(self, ab: AttentionBlock, x, qkv, a, out):
        B, H, W, Cc = x.shape
        T = H * W
        dout, have = self._grad_of(out)
        assert have
        dout16 = self._bias_grad(dout, ab.proj_out.bias, cast="dy")
        a16 = self._cast16(a, kind="x16")
        self._wgrad(a16, dout16, dout, ab.proj_out.weight, 1, 0)
        da = self._buf(f"attn.da.{B}x{T}x{Cc}", (B, H, W, Cc))
        self._dgrad(ab.proj_out, dout16, da)
        dqkv = self._buf(f"attn.dqkv.{B}x{T}x{Cc}", (B, H, W, 3 * Cc))
        ops.attn_legacy_bwd(qkv.view(B, T, 3 * Cc), da.view(B, T, Cc), dqkv.view(B, T, 3 * Cc), ab.num_heads)
        dqkv16 = self._bias_grad(dqkv, ab.qkv.bias, cast="dy3")
        n16 = self._norm16(ab.norm, 0, x)
        self._wgrad(n16, dqkv16, dqkv, ab.qkv.weight, 1, 0)
        dn = self._buf(f"attn.dn.{B}x{T}x{Cc}", (B, H, W, Cc))
        self._dgrad(ab.qkv, dqkv16, dn)
        self._gn_bwd(ab.norm, 0, x, None, dn, dout)

    def _st_bwd(self, st, x, saved, out):
        """SpatialTransformer (ldm/modules/attention.py:218-261, routed without context: both attentions are self-attentions) from the stage
        tensors its training-mode forward kept (SpatialTransformer.run(save=...)). Every Linear is a 1x1 convolution over the token rows
        (dgrad / wgrad kernels of the convolutions), the attentions run the QKVAttentionLegacy backward on the stacked [head][q | k | v] rows
        (the forward's own form), LayerNorm and GEGLU have their backward kernels (stedm_ln_bwd, stedm_geglu_bwd)."""
        B, H, W, C = x.shape
        T, inner, heads = H * W, st.inner_dim, st.n_heads
        M = B * T
        d = inner // heads
        bp = self.bprec
        v = lambda t, n: t.view(B, H, W, n)             # token rows [M, n] as an NHWC tensor for the helpers
        dout, have = self._grad_of(out)
        assert have
        # ---- proj_out: out = conv1x1(tokens) + bias + x
        dout16 = self._bias_grad(dout, st.proj_out.bias, cast="dy")
        self._wgrad(self._cast16(v(saved["yL"], inner), kind="x16"), dout16, dout, st.proj_out.weight, 1, 0)
        dy = self._buf(f"st.dy0.{M}x{inner}", (B, H, W, inner))
        dy2 = self._buf(f"st.dy1.{M}x{inner}", (B, H, W, inner))
        self._dgrad(st.proj_out, dout16, dy)

        lnws = self._buf(f"st.lnws.{M}x{inner}", (ops.ln_bwd_ws_floats(M, inner),))

        def ln_planes(y_in, norm):
            hi, lo = self._planes("st.ln16", (B, H, W, inner))
            ops.ln_apply16(y_in, norm.weight, norm.bias, norm.eps, hi, lo, bp)
            return hi, lo

        for i in reversed(range(len(st.transformer_blocks))):
            blk = st.transformer_blocks[i]
            # ---- feed-forward: y_out = y_in + W_out GEGLU(W_proj LN3(y_in) + b_proj) + b_out
            y_in, g, _ = saved[f"{i}.ff"]
            lin_out, lin_proj = blk.ff.net[2], blk.ff.net[0].proj
            dy16 = self._bias_grad(dy, lin_out.bias, cast="dy")
            h16 = self._planes("st.h16", (B, H, W, 4 * inner))
            ops.geglu16(g, h16[0], h16[1], bp)
            self._wgrad(h16, dy16, dy, lin_out.weight, 1, 0)
            dh = self._buf(f"st.dh.{M}", (B, H, W, 4 * inner))
            self._dgrad(lin_out, dy16, dh)
            dg = self._buf(f"st.dg.{M}", (B, H, W, 8 * inner))
            ops.geglu_bwd(g, dh.view(M, 4 * inner), dg.view(M, 8 * inner))
            dg16 = self._bias_grad(dg, lin_proj.bias, cast="dy8")
            self._wgrad(ln_planes(y_in, blk.norm3), dg16, dg, lin_proj.weight, 1, 0)
            dln = self._buf(f"st.dln.{M}", (B, H, W, inner))
            self._dgrad(lin_proj, dg16, dln)
            ops.ln_bwd(y_in, dln.view(M, inner), blk.norm3.weight, blk.norm3.eps, dy2.view(M, inner), self._param_grad(blk.norm3.weight),
                       self._param_grad(blk.norm3.bias), add=dy.view(M, inner), ws=lnws)
            dy, dy2 = dy2, dy
            # ---- the two self-attentions: y_out = y_in + W_out attention(W_qkv LN(y_in)) + b_out
            for nm, norm in (("attn2", blk.norm2), ("attn1", blk.norm1)):
                at = getattr(blk, nm)
                y_in, qkv, att, _ = saved[f"{i}.{nm}"]
                lin_o = at.to_out[0]
                dy16 = self._bias_grad(dy, lin_o.bias, cast="dy")
                self._wgrad(self._cast16(v(att, inner), kind="x16"), dy16, dy, lin_o.weight, 1, 0)
                datt = self._buf(f"st.datt.{M}", (B, H, W, inner))
                self._dgrad(lin_o, dy16, datt)
                dqkv = self._buf(f"st.dqkv.{M}", (B, H, W, 3 * inner))
                ops.attn_legacy_bwd(qkv.view(B, T, 3 * inner), datt.view(B, T, inner), dqkv.view(B, T, 3 * inner), heads)
                dqkv16 = self._cast16(dqkv, kind="dy3")
                # the stacked filter rows [head][q | k | v][d] (SpatialTransformer.pack): one weight gradient, scattered to the three Linears
                gq = self._buf(f"st.gqkv.{inner}", (3 * inner, inner))
                self._wgrad(ln_planes(y_in, norm), dqkv16, dqkv, None, 1, 0, grad=gq)
                g4 = gq.view(heads, 3, d, inner)
                for j, lin in enumerate((at.to_q, at.to_k, at.to_v)):
                    self._param_grad(lin.weight).copy_(g4[:, j].reshape(inner, inner))
                holder = SimpleNamespace(weight=torch.stack([at.to_q.weight.detach().float().reshape(heads, d, -1),
                                                             at.to_k.weight.detach().float().reshape(heads, d, -1),
                                                             at.to_v.weight.detach().float().reshape(heads, d, -1)], dim=1).reshape(3 * inner, -1))
                self._st_holders.append(holder)           # (keeps id(holder) unique for the per-step dgrad pack cache)
                self._dgrad(holder, dqkv16, dln)
                ops.ln_bwd(y_in, dln.view(M, inner), norm.weight, norm.eps, dy2.view(M, inner), self._param_grad(norm.weight),
                           self._param_grad(norm.bias), add=dy.view(M, inner), ws=lnws)
                dy, dy2 = dy2, dy
        # ---- proj_in: tokens = conv1x1(GroupNorm(x)) + bias
        dy16 = self._bias_grad(dy, st.proj_in.bias, cast="dy")
        self._wgrad(self._norm16(st.norm, 0, x), dy16, dy, st.proj_in.weight, 1, 0)
        dn = self._buf(f"st.dn.{M}", (B, H, W, C))
        self._dgrad(st.proj_in, dy16, dn)
        self._gn_bwd(st.norm, 0, x, None, dn, dout)

    def _up_bwd(self, layer: Upsample, hin, out):
        B, H, W, Cc = hin.shape
        dout, have = self._grad_of(out)
        assert have
        dout16 = self._bias_grad(dout, layer.conv.bias, cast="dy")
        src16 = self._cast16(hin, kind="x16")
        if self.direct_wgrad and self.bprec.npass == 1 and ops.wgrad3x3_plan(B, 2 * H, 2 * W, Cc, out.shape[-1]) > 0:
            # the direct kernel reads plain NHWC planes: materialise the nearest-2x plane once (16-bit copy, layout only)
            up = self._buf(f"up.x16.{B}x{H}x{W}x{Cc}", (B, 2 * H, 2 * W, Cc), torch.int16)
            up.view(B, H, 2, W, 2, Cc).copy_(src16[0].view(B, H, 1, W, 1, Cc).expand(B, H, 2, W, 2, Cc))
            self._wgrad((up, None), dout16, dout, layer.conv.weight, 3, 0)
        else:
            self._wgrad(src16, dout16, dout, layer.conv.weight, 3, 1)
        tmp = self._buf(f"up.t.{B}x{H}x{W}x{Cc}", (B, 2 * H, 2 * W, Cc))
        self._dgrad(layer.conv, dout16, tmp)
        g, acc = self._grad_of(hin)
        ops.sum2x2(tmp, g, acc)

    def _down_bwd(self, layer: Downsample, hin, out):
        B, H, W, Cc = hin.shape
        co = out.shape[-1]
        dout, have = self._grad_of(out)
        assert have
        dout16 = self._bias_grad(dout, layer.op.bias, cast="dy")
        src16 = self._cast16(hin, kind="x16")
        self._wgrad(src16, dout16, dout, layer.op.weight, 3, 2)
        z16 = self._planes("z16", (B, H, W, co))
        ops.zero_insert16(dout, z16[0], z16[1], self.bprec)
        g, acc = self._grad_of(hin)
        self._dgrad(layer.op, z16, g, accumulate=acc)

    def _colsum(self, mat: torch.Tensor, dst: torch.Tensor) -> None:
        """dst[n] = sum_b mat[b][n]"""
        B, N = mat.shape
        cs = self._buf(f"colsum.{B}x{N}", (1, ops.gn_chan_nslab(B), N, 2))
        ops.gn_chan_stats(mat.view(1, B, 1, N), cs)
        ops.chan_sum_fold(cs, None, 0, dst, False)

    def _emb_bwd(self, B: int, ted: int) -> torch.Tensor:
        """backward of the embedding paths: emb_layers of every ResBlock (one concatenated Linear, openaimodel.py:231-237),
        time_embed (:529-534; the sinusoid has no parameters), and the style block's emb_layers on the context vector."""
        m = self.m
        c = m._consts
        timesteps, emb, ctx = self.tape_emb
        mc = m.model_channels
        # ---- ResBlock emb_layers: E = silu(emb) @ Wcat^T + b
        S = ops.silu(emb, self._buf("emb.S", (B, ted)))
        self._param_grad(m._emb_layout[0][0].emb_layers[1].weight)
        ops.gemm_f32(self.dE, True, S, False, self._dWcat)       # the emb_layers weight gradients are row blocks of this matrix
        if self._touched is not None:
            self._touched.extend(rb.emb_layers[1].weight for rb, _ in m._emb_layout)
        gws = self._buf("emb.gws", (64 * B * ted,))
        dS = ops.gemm_f32(self.dE, False, c["emb_w"], False, self._buf("emb.dS", (B, ted)), ws=gws)    # dE @ Wcat: few rows, long K -> split-K
        demb = ops.silu(emb, self._buf("emb.demb", (B, ted)), dy=dS)
        # ---- time_embed: emb = silu(te @ W0^T + b0) @ W2^T + b2
        l0, l2 = m.time_embed[0], m.time_embed[2]
        te = m._buf("emb_ws", (B * (mc + ted),))[:B * mc].view(B, mc)
        u = ops.linear(te, c["te_w0t"], c["te_b0"], self._buf("emb.u", (B, ted)))
        h1 = ops.silu(u, self._buf("emb.h1", (B, ted)))
        ops.gemm_f32(demb, True, h1, False, self._param_grad(l2.weight))
        self._colsum(demb, self._param_grad(l2.bias))
        dh1 = ops.gemm_f32(demb, False, l2.weight.detach(), False, self._buf("emb.dh1", (B, ted)))
        du = ops.silu(u, self._buf("emb.du", (B, ted)), dy=dh1)
        ops.gemm_f32(du, True, te, False, self._param_grad(l0.weight))
        self._colsum(du, self._param_grad(l0.bias))
        # ---- style block: Es = silu(ctx) @ Ws^T + bs  (ResBlockStyle: emb = context, openaimodel.py:291-297)
        srb = m.middle_block[1].block
        lin = srb.emb_layers[1]
        ctx = ctx.float().contiguous()
        Sc = ops.silu(ctx, self._buf("emb.Sc", (B, ted)))
        ops.gemm_f32(self.dEs, True, Sc, False, self._param_grad(lin.weight))
        dSc = ops.gemm_f32(self.dEs, False, lin.weight.detach(), False, self._buf("emb.dSc", (B, ted)))
        return ops.silu(ctx, torch.empty_like(ctx), dy=dSc)

    # ------------------------------------------------------------------------------------------------ loss + optimizer
    @torch.no_grad()
    def loss_and_backward(self, x, c_concat, t, context, target, on_bucket=None, sched=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """L1 loss of ddpm.py:1030-1040 on the eps prediction, then the full backward. -> (loss [1] device tensor, dx, dcontext)"""
        pred = self.forward(x, c_concat, t, context)
        loss = self._buf("loss", (1,))
        dpred = self._buf(f"dpred.{tuple(pred.shape)}", tuple(pred.shape))
        ops.l1_loss(pred, target.float().contiguous(), dpred, self._buf("loss.ws", (1024,), torch.float64), loss)
        dx, dctx = self.backward(dpred, on_bucket=on_bucket, sched=sched)
        return loss, dx, dctx

    def _chunks(self, params, dev):
        ct, co = [], []
        for i, p in enumerate(params):
            for o in range(0, p.numel(), 4096):
                ct.append(i); co.append(o)
        return torch.tensor(ct, dtype=torch.int32, device=dev), torch.tensor(co, dtype=torch.int64, device=dev)

    def _build_ema(self):
        """EMA shadows of the U-Net's parameters (LitEma is built over `model`, ddpm.py:86-88; the cond stage has none) and the
        pointer table the EMA-only kernel walks. Independent of the optimizer state: a training loop that keeps torch.optim.AdamW
        (the autograd bridge of latent_diffusion.py) still gets on_train_batch_end's EMA from here."""
        import numpy as np
        if getattr(self, "grad_arena", None) is None:
            self._alloc_grads()
        params = list(self._arena_params)
        dev = params[0].device
        ema = [p.detach().clone() if i < self._n_unet_params else None for i, p in enumerate(params)] if self.ema_decay is not None else None
        tab = np.zeros((len(params), 6), dtype=np.int64)
        for i, p in enumerate(params):
            assert p.is_contiguous()
            tab[i] = (p.data_ptr(), 0, 0, 0, ema[i].data_ptr() if ema is not None and ema[i] is not None else 0, p.numel())
        ct, co = self._chunks(params, dev)
        self._ema = {"ema": ema, "params": params, "tab_np": tab, "table": torch.from_numpy(tab).to(dev), "ct": ct, "co": co}
        if getattr(self, "_ema_resume", None) is not None:
            self._apply_ema_resume()

    def _build_opt(self):
        import numpy as np
        if getattr(self, "_ema", None) is None:
            self._build_ema()
        params = list(self._arena_params)
        dev = params[0].device
        st = {"m": [torch.zeros_like(p, dtype=torch.float32) for p in params], "v": [torch.zeros_like(p, dtype=torch.float32) for p in params],
              "ema": self._ema["ema"], "params": params}
        tab = self._ema["tab_np"].copy()
        for i, p in enumerate(params):
            g = self.grad_arena[self._arena_off[i]:self._arena_off[i] + p.numel()]
            tab[i, 1:4] = (g.data_ptr(), st["m"][i].data_ptr(), st["v"][i].data_ptr())
        st["table"] = torch.from_numpy(tab).to(dev)
        st["ct"], st["co"] = self._ema["ct"], self._ema["co"]
        self._opt = st
        if getattr(self, "_opt_resume", None) is not None:
            self._apply_opt_resume()

    def _fused_opt(self):
        """Descriptor table of stedm_adamw_ema_pack: every arena parameter that is a convolution weight with fragment-order packs recorded
        in the forward's plan (UNetModel._plan) and / or the backward's (self._dplan), plus the chunk lists of the remaining tensors for the
        plain kernel. None while no plan exists (before the first forward + backward) or nothing qualifies. Rebuilt when a plan changes."""
        import struct
        m = self.m
        plans = [pl for pl in (getattr(m, "_plan", None), getattr(self, "_dplan", None)) if pl is not None and pl.items]
        if not plans or (len(plans) == 2 and plans[0] is plans[1]):
            return None
        key = tuple((id(pl), len(pl.items)) for pl in plans)
        fu = getattr(self, "_fused", None)
        if fu is not None and fu["key"] == key:
            return fu if fu["n"] else None
        st = self._opt
        params = st["params"]
        tab = st["table"].cpu().numpy()
        by_ptr = {}
        for pl in plans:
            if pl.prec.npass != 1:
                continue
            for j, (w, sn, sc, flip, cout, cin, taps, m16, out) in enumerate(pl.items):
                by_ptr.setdefault(w.data_ptr(), []).append((pl, j, sn, sc, flip, cout, cin, taps, m16, out))
        rec, blk, fused_idx, used = [], 0, set(), {id(pl): [] for pl in plans}
        rec_parts = []
        prow, pciw = ops.adamw_ema_pack_piece()
        for i, p in enumerate(params):
            its = by_ptr.get(p.data_ptr())
            if not its or p.dim() < 3 or not p.is_contiguous():
                continue
            co, ci = p.shape[0], p.shape[1]
            taps = p.numel() // (co * ci)
            if co % prow or ci % pciw or taps not in (1, 9) or len(its) > 4:
                continue
            outs = []
            for (pl, j, sn, sc, flip, pcout, pcin, ptaps, m16, out) in its:
                straight = (sn, sc, pcout, pcin, flip) == (ci * taps, taps, co, ci, 0)
                transposed = (sn, sc, pcout, pcin, flip) == (taps, ci * taps, ci, co, 1)
                if ptaps != taps or not (straight or transposed):
                    outs = None
                    break
                outs.append((pl, j, struct.pack("<Qiiii", out.data_ptr(), int(transposed), int(flip), int(m16), int(pl.prec.mm_dtype == F16))))
            if not outs:
                continue
            body = b"".join(o[2] for o in outs) + b"\0" * (24 * (4 - len(outs)))
            rec.append(struct.pack("<QQQQQiiiiii", int(tab[i, 0]), int(tab[i, 1]), int(tab[i, 2]), int(tab[i, 3]), int(tab[i, 4]), co, ci, taps, blk,
                                   len(outs), 0) + body)
            rec_parts.append((i, tuple(int(tab[i, c_]) for c_ in range(5)), co, ci, taps, len(outs), body, (co // prow) * (ci // pciw)))
            blk += (co // prow) * (ci // pciw)
            fused_idx.add(i)
            for (pl, j, _) in outs:
                used[id(pl)].append(j)
        dev = params[0].device
        fu = {"key": key, "n": len(rec), "blocks": blk, "plans": plans}
        if rec:
            fu["descs"] = torch.frombuffer(bytearray(b"".join(rec)), dtype=torch.uint8).to(dev)
            rest = [(i, p) for i, p in enumerate(params) if i not in fused_idx]
            ct, co_ = [], []
            for i, p in rest:
                for o in range(0, p.numel(), 4096):
                    ct.append(i); co_.append(o)
            fu["ct"] = torch.tensor(ct, dtype=torch.int32, device=dev)
            fu["co"] = torch.tensor(co_, dtype=torch.int64, device=dev)
            for pl in plans:
                pl.set_fused(used[id(pl)])
            # the same two launches per run of the overlapped optimizer's schedule (own tables, block offsets from 0)
            osched = self._opt_sched()
            fu["runs"] = []
            for b in range(len(osched.bounds)):
                mine = [r for r in rec_parts if osched.param_bucket[r[0]] == b]
                rb, nb_ = [], 0
                for (i, t5, co, ci, taps, nouts, body, nblk) in mine:
                    rb.append(struct.pack("<QQQQQiiiiii", *t5, co, ci, taps, nb_, nouts, 0) + body)
                    nb_ += nblk
                cti, coi = [], []
                for i, p in rest:
                    if osched.param_bucket[i] == b:
                        for o in range(0, p.numel(), 4096):
                            cti.append(i); coi.append(o)
                fu["runs"].append({"descs": torch.frombuffer(bytearray(b"".join(rb)), dtype=torch.uint8).to(dev) if rb else None, "n": len(rb), "blocks": nb_,
                                   "ct": torch.tensor(cti, dtype=torch.int32, device=dev), "co": torch.tensor(coi, dtype=torch.int64, device=dev)})
        self._fused = fu
        return fu if fu["n"] else None

    @torch.no_grad()
    def all_reduce_grads(self, group=None, bucket_mb: int = 256) -> int:
        """Data-parallel training (train_diff.py runs Lightning DDP): sum the gradient arena over the ranks in a few large buckets
        (xGMI rings are per-link bound: few, large collectives); the 1/world average is folded into the optimizer kernel.
        Returns the world size."""
        from .parallel import all_reduce_bounds
        world = all_reduce_bounds(self.grad_arena, self._sched.bounds, group)
        self._grad_scale = 1.0 / world
        return world

    def _next_ema_decay(self) -> float:
        """ema.py:28-31: the counter is incremented first, decay = min(decay, (1 + n) / (10 + n))."""
        self.ema_updates += 1
        return min(self.ema_decay, (1 + self.ema_updates) / (10 + self.ema_updates))

    def _opt_sched(self):
        """Runs of the arena (cut at parameter boundaries, about opt_bucket_mb each) whose optimizer pass starts inside the backward."""
        if getattr(self, "_osched", None) is None:
            from .parallel import BucketSchedule
            order = self._arena_params
            self._osched = BucketSchedule(self._arena_off, [p.numel() for p in order], self.opt_bucket_mb * (1 << 20) // 4, tail_from=self._n_unet_params,
                                          align=4, total=self.grad_arena.numel())
        return self._osched

    def _opt_run(self, fu, b: int, decay: float, gs: float) -> None:
        """the optimizer pass of run b of _opt_sched() on the current stream"""
        st, r = self._opt, fu["runs"][b]
        if r["n"]:
            self._adamw_pack(r["descs"], r["n"], r["blocks"], decay, gs)
        if r["ct"].numel():
            self._adamw(st["table"], r["ct"], r["co"], decay, gs)

    def _adamw(self, table, ct, co, decay: float, gs: float) -> None:
        cs = getattr(self, "_cap_sched", None)
        if cs is not None:       # inside capture_step(): bias corrections / EMA decay / lr of the replayed step come from the device schedule
            ops.adamw_ema_sched(table, ct, co, self.betas[0], self.betas[1], self.eps, self.wd, cs[0], cs[1], grad_scale=gs)
        else:
            ops.adamw_ema(table, ct, co, self.lr, self.betas[0], self.betas[1], self.eps, self.wd, self.step_count, decay, grad_scale=gs)

    def _adamw_pack(self, descs, n: int, blocks: int, decay: float, gs: float) -> None:
        cs = getattr(self, "_cap_sched", None)
        if cs is not None:
            ops.adamw_ema_pack_sched(descs, n, blocks, self.betas[0], self.betas[1], self.eps, self.wd, cs[0], cs[1], grad_scale=gs)
        else:
            ops.adamw_ema_pack(descs, n, blocks, self.lr, self.betas[0], self.betas[1], self.eps, self.wd, self.step_count, decay, gs)

    def _after_optimizer(self, fu) -> None:
        self.m.invalidate()      # parameters changed through raw pointers: repack on the next forward
        if fu is not None:
            # ... except what this pass wrote itself: valid while neither a parameter's version nor the model's value generation
            # (UNetModel.invalidate(): EMA swap, checkpoint load, edits through .data / raw pointers) moves
            token = self.m.freshness_token()
            for plan in fu["plans"]:
                plan.mark_fresh(token)
        self._grads_ready = False

    @torch.no_grad()
    def optimizer_step(self) -> None:
        """AdamW over every parameter + EMA shadow update (ema.py:25-44: decay = min(decay, (1+n)/(10+n))) in one launch; gradients
        are read from the flat arena."""
        assert self._grads_ready, "optimizer_step() needs gradients from backward()"
        if self._opt is None:
            self._build_opt()
        st = self._opt
        self.step_count += 1
        decay = self._next_ema_decay() if self.ema_decay is not None else 0.0
        gs = getattr(self, "_grad_scale", 1.0)
        fu = self._fused_opt() if self.fuse_packs else None
        if fu is None:
            self._adamw(st["table"], st["ct"], st["co"], decay, gs)
        else:
            # convolution weights with fragment-order packs: the optimizer pass writes the packs of the next forward / backward itself
            self._adamw_pack(fu["descs"], fu["n"], fu["blocks"], decay, gs)
            if fu["ct"].numel():
                self._adamw(st["table"], fu["ct"], fu["co"], decay, gs)
        self._after_optimizer(fu)

    @torch.no_grad()
    def ema_step(self) -> None:
        """LitEma.forward alone: what `on_train_batch_end` (ddpm.py:369-371) runs after EVERY micro-batch, also those of an
        accumulation window that did not step the optimizer (the shadows then move towards unchanged parameters)."""
        if self.ema_decay is None:
            return
        if getattr(self, "_ema", None) is None:
            self._build_ema()
        e = self._ema
        ops.ema_update(e["table"], e["ct"], e["co"], self._next_ema_decay())

    def ema_parameters(self) -> Optional[List[torch.Tensor]]:
        return None if getattr(self, "_ema", None) is None else self._ema["ema"]

    def ema_named(self) -> Optional[Dict[str, torch.Tensor]]:
        """EMA shadows by U-Net parameter name (None before the shadows exist or without EMA)."""
        if getattr(self, "_ema", None) is None or self._ema["ema"] is None:
            return None
        names = {id(p): n for n, p in self.m.named_parameters()}
        return {names[id(p)]: e for p, e in zip(self._ema["params"], self._ema["ema"]) if e is not None}

    @torch.no_grad()
    def load_ema(self, shadows: Dict[str, torch.Tensor], num_updates: int) -> None:
        """Resume the EMA of a checkpoint (LitEma buffers by U-Net parameter name + its update counter)."""
        self._ema_resume = (dict(shadows), int(num_updates))
        if getattr(self, "_ema", None) is not None:
            self._apply_ema_resume()

    def _apply_ema_resume(self) -> None:
        shadows, n = self._ema_resume
        named = self.ema_named()
        if named is not None:
            for name, e in named.items():
                if name in shadows:
                    e.copy_(shadows[name].to(e.device, e.dtype))
        self.ema_updates = n
        self._ema_resume = None

    # ------------------------------------------------------------------------------------------------ optimizer state (checkpoints)
    def _torch_param_order(self) -> List[nn.Parameter]:
        """the order `configure_optimizers` hands the parameters to torch.optim.AdamW (modules/ldm_diffusion.py:224-234):
        model.model.parameters() then cond_stage_model's; optimizer state dicts index them by that position"""
        return list(self.m.parameters()) + list(self.extra_params)

    @torch.no_grad()
    def optimizer_state_dict(self) -> dict:
        """AdamW state in `torch.optim.AdamW.state_dict()` layout (what a Lightning checkpoint keeps under `optimizer_states[0]`):
        {"state": {i: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [...]}, i in the reference's parameter order."""
        if self._opt is None:
            self._build_opt()
        pos = {id(p): i for i, p in enumerate(self._opt["params"])}
        state = {}
        order = self._torch_param_order()
        for i, p in enumerate(order):
            j = pos[id(p)]
            state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self._opt["m"][j].detach().clone(),
                        "exp_avg_sq": self._opt["v"][j].detach().clone()}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.wd, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(order)))}
        return {"state": state, "param_groups": [group]}

    @torch.no_grad()
    def load_optimizer_state_dict(self, sd: dict) -> None:
        """Resume AdamW from a state dict in torch's layout (see optimizer_state_dict): both moments and the step counter (bias
        correction continues instead of restarting); lr / betas / eps / weight_decay of the first param group are taken over."""
        self._opt_resume = sd
        if self._opt is not None:
            self._apply_opt_resume()

    def _apply_opt_resume(self) -> None:
        sd, self._opt_resume = self._opt_resume, None
        pos = {id(p): i for i, p in enumerate(self._opt["params"])}
        order = self._torch_param_order()
        steps = set()
        for i, p in enumerate(order):
            st = sd["state"].get(i, sd["state"].get(str(i)))
            if st is None:
                continue
            j = pos[id(p)]
            self._opt["m"][j].copy_(st["exp_avg"].to(self._opt["m"][j].device, torch.float32).view_as(self._opt["m"][j]))
            self._opt["v"][j].copy_(st["exp_avg_sq"].to(self._opt["v"][j].device, torch.float32).view_as(self._opt["v"][j]))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"optimizer state with per-parameter step counts {sorted(steps)}: the fused AdamW kernel keeps one counter")
        if steps:
            self.step_count = steps.pop()
        g = (sd.get("param_groups") or [{}])[0]
        self.lr = float(g.get("lr", self.lr))
        self.betas = tuple(float(b) for b in g.get("betas", self.betas))
        self.eps, self.wd = float(g.get("eps", self.eps)), float(g.get("weight_decay", self.wd))

    # ------------------------------------------------------------------------------------------------ autograd bridge support
    @torch.no_grad()
    def publish_grads(self, scale: float, accumulate: bool) -> None:
        """acc_arena = scale * grad_arena (+ acc_arena when `accumulate`) and every parameter's `.grad` becomes a view of acc_arena:
        autograd's accumulate-into-.grad semantics for the bridge of latent_diffusion.py (`loss.backward()` called several times
        between two `optimizer.zero_grad()`), while the backward kernels keep overwriting grad_arena."""
        if self._pub_arena is None:
            self._pub_arena = torch.zeros_like(self.grad_arena)
            self._pub_views = [self._pub_arena[o:o + p.numel()].view(p.shape) for o, p in zip(self._arena_off, self._arena_params)]
        ops.axpby(self.grad_arena, self._pub_arena, float(scale), 1.0 if accumulate else 0.0)
        for p, v in zip(self._arena_params, self._pub_views):
            p.grad = v

    def internal_grads(self) -> bool:
        """Point every `.grad` back at the kernels' arena before a backward pass; returns whether the caller-visible gradients held a
        contribution (they were the published views: no zero_grad(set_to_none=True) since the last publish_grads)."""
        if getattr(self, "grad_arena", None) is None:
            self._alloc_grads()
            return False
        if self._pub_arena is None:
            if any(p.grad is None for p in self._arena_params):
                for p, v in zip(self._arena_params, self._int_views):
                    p.grad = v
            return False
        had = all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(self._arena_params, self._pub_views))
        for p, v in zip(self._arena_params, self._int_views):
            p.grad = v
        return had

    @torch.no_grad()
    def train_step(self, x, c_concat, t, context, target, group=None, after_backward=None) -> torch.Tensor:
        """One micro-batch: loss + backward; every `accumulate_grad_batches`-th call also all-reduces (data parallel) and steps the
        optimizer on the mean of the accumulated gradients (Lightning divides the loss by the accumulation count); LitEma's update
        runs after every micro-batch. after_backward(dx, dcontext): fills the gradients of `extra_params` (cond stage)."""
        import torch.distributed as dist
        k = self.accumulate_grad_batches
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        last = self._micro + 1 >= k
        works = []
        on_bucket = None
        if getattr(self, "grad_arena", None) is None:
            self.m._prepare()                # (the arena's layout follows the packed embedding-Linear order; cached for the forward)
            self._alloc_grads()              # before deciding: the FIRST step overlaps its all-reduce like every later one
        if multi and last and self.overlap_all_reduce:
            # the collective of a bucket starts the moment the backward has produced its last gradient (the all-reduce then runs beside
            # the remaining dgrad / wgrad kernels); with accumulation the bucket is first folded into the running mean
            def on_bucket(b):
                lo, hi = self._sched.bounds[b]
                if b == len(self._sched.bounds) - 1 and self.extra_params:
                    return                      # the cond stage's gradients arrive after the backward: reduced below
                src = self.grad_arena
                if k > 1:
                    ops.axpby(self.grad_arena[lo:hi], self._acc_arena[lo:hi], 1.0 / k, 1.0 if self._micro > 0 else 0.0)
                    src = self._acc_arena
                works.append(dist.all_reduce(src[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True))
                self.overlap_fires += 1          # collectives issued from inside the backward (tests assert the overlap path really ran)
        if k > 1 and getattr(self, "_acc_arena", None) is None:
            self._acc_arena = torch.empty_like(self.grad_arena)
        fu = None
        if self.overlap_optimizer and not multi and k == 1 and self.fuse_packs and self._opt is not None:
            fu = self._fused_opt()      # (None until the forward's and the backward's pack plans exist: the first steps run the plain order)
        if fu is not None and fu.get("runs"):
            osched = self._opt_sched()
            self.step_count += 1
            decay = self._next_ema_decay() if self.ema_decay is not None else 0.0
            gs = getattr(self, "_grad_scale", 1.0)
            if getattr(self, "_opt_stream", None) is None:
                self._opt_stream = torch.cuda.Stream()
            side, main = self._opt_stream, torch.cuda.current_stream()

            def on_run(b):
                ev = torch.cuda.Event()
                ev.record(main)              # every gradient of run b has been written by kernels queued so far
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    self._opt_run(fu, b, decay, gs)
                self.overlap_opt_fires += 1

            loss, dx, dctx = self.loss_and_backward(x, c_concat, t, context, target, on_bucket=on_run, sched=osched)
            if after_backward is not None:
                after_backward(dx, dctx)
            self._grads_ready = True
            main.wait_stream(side)
            for b in range(len(osched.bounds)):      # runs that did not complete inside the backward (the cond stage's tail, untouched parameters)
                if not osched.fired[b]:
                    self._opt_run(fu, b, decay, gs)
            self._after_optimizer(fu)
            return loss
        loss, dx, dctx = self.loss_and_backward(x, c_concat, t, context, target, on_bucket=on_bucket)
        if after_backward is not None:
            after_backward(dx, dctx)
        if k > 1 and getattr(self, "_acc_arena", None) is None:
            self._acc_arena = torch.empty_like(self.grad_arena)
        if on_bucket is not None:
            nb = len(self._sched.bounds)
            rest = [b for b in range(nb) if not self._sched.fired[b] or (b == nb - 1 and self.extra_params)]
            for b in rest:                       # buckets that did not complete inside the backward (the cond stage's tail)
                lo, hi = self._sched.bounds[b]
                src = self.grad_arena
                if k > 1:
                    ops.axpby(self.grad_arena[lo:hi], self._acc_arena[lo:hi], 1.0 / k, 1.0 if self._micro > 0 else 0.0)
                    src = self._acc_arena
                works.append(dist.all_reduce(src[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True))
            for w in works:
                w.wait()
            self._grad_scale = 1.0 / dist.get_world_size(group)
            if k > 1:
                ops.axpby(self._acc_arena, self.grad_arena, 1.0, 0.0)
                self._micro = 0
            self.optimizer_step()
            return loss
        if k > 1:
            ops.axpby(self.grad_arena, self._acc_arena, 1.0 / k, 1.0 if self._micro > 0 else 0.0)
            self._micro += 1
            if self._micro < k:
                self._grads_ready = False
                self.ema_step()          # on_train_batch_end runs LitEma after every micro-batch (ddpm.py:369-371)
                return loss
            ops.axpby(self._acc_arena, self.grad_arena, 1.0, 0.0)      # the optimizer table points at the gradient arena
            self._micro = 0
        if multi:
            self.all_reduce_grads(group)
        self.optimizer_step()
        return loss

    # ------------------------------------------------------------------------------------------------ captured step (hipGraph replay)
    GRAPH_WARMUP = 3          # eager steps before the capture: the forward's and the backward's pack plans and the fused optimizer table exist from the 3rd
    GRAPH_WINDOW = 4096       # steps per upload of the device schedule

    def _sched_rows(self, first_step: int, first_ema: int, n: int):
        """{1 - beta1^s, sqrtf(1 - beta2^s), LitEma decay, lr} for steps first_step .. first_step + n - 1 with the arithmetic of stedm_adamw_ema
        (bwd.hip: betas arrive as C floats, the powers are taken in double) and of _next_ema_decay — a replayed step uses the very scalars the
        eager step would have been launched with"""
        import numpy as np
        b1, b2 = float(np.float32(self.betas[0])), float(np.float32(self.betas[1]))
        rows = np.empty((n, 4), np.float32)
        for i in range(n):
            s_ = first_step + i
            rows[i, 0] = np.float32(1.0 - b1 ** s_)
            rows[i, 1] = np.sqrt(np.float32(1.0 - b2 ** s_))
            e = first_ema + i
            rows[i, 2] = np.float32(min(self.ema_decay, (1 + e) / (10 + e))) if self.ema_decay is not None else 0.0
            rows[i, 3] = np.float32(self.lr)
        return rows

    def _graph_window(self, g) -> None:
        """(re)fill the device schedule so that row 1 is the NEXT step (the captured step advances the index first) and rewind the index"""
        n = self.GRAPH_WINDOW
        rows = self._sched_rows(self.step_count, self.ema_updates, n)      # row 0 = the step already taken (never read), row k = step_count + k
        g["sched"].copy_(torch.from_numpy(rows), non_blocking=False)
        g["idx"].zero_()
        g["base"], g["lr"] = self.step_count, self.lr

    @torch.no_grad()
    def train_step_graphed(self, x, c_concat, t, context, target) -> torch.Tensor:
        """train_step() of one rank without gradient accumulation as ONE hipGraph launch (ddpm.py:345-371: training_step, optimizer.step,
        on_train_batch_end — shape-static). The first GRAPH_WARMUP calls run eagerly (pack plans and the fused optimizer table are built by
        them), the next call captures the step — forward, L1 loss, backward, AdamW + EMA + re-pack, about 790 launches — and every call from
        then on copies the batch into the captured input buffers and replays it. What changes between steps (AdamW's bias corrections,
        LitEma's decay, the learning rate) is read by the optimizer kernels from a device schedule the host fills for GRAPH_WINDOW steps at
        a time (stedm_adamw_ema_sched); the step index advances inside the graph. Same kernels on the same values as the eager step: the
        parameters after k replayed steps equal those after k eager steps bit for bit (tests/test_gpu_train.py). The graph is dropped (and
        re-captured after GRAPH_WARMUP eager steps) when the shapes change or anything outside this method touched the parameters (EMA swap,
        checkpoint load, an eager step). Returns the loss [1] (device tensor, overwritten by the next step)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            raise NotImplementedError("train_step_graphed: one rank only (the bucketed all-reduce of train_step() is issued by the host)")
        if self.accumulate_grad_batches != 1 or self.extra_params:
            raise NotImplementedError("train_step_graphed: no gradient accumulation, no parameters outside the U-Net (their gradients are filled by a "
                                      "host callback between the backward and the optimizer)")
        if x.device.type != "cuda":
            raise RuntimeError("train_step_graphed needs the HIP path (a CUDA/HIP device tensor)")
        args = (x, c_concat, t, context, target)
        key = tuple(None if a is None else (tuple(a.shape), a.dtype) for a in args)
        g = getattr(self, "_graph", None)
        if g is not None and (g["key"] != key or g["token"] != self.m.freshness_token() or g["host_step"] != self.step_count):
            g = self._graph = None                    # someone else moved the parameters or the counters: the captured plan is stale
            self._graph_eager = 0
        if g is None:
            if getattr(self, "_graph_eager", 0) < self.GRAPH_WARMUP or getattr(self, "_graph_key", None) != key:
                if getattr(self, "_graph_key", None) != key:
                    self._graph_key, self._graph_eager = key, 0
                self._graph_eager += 1
                return self.train_step(x, c_concat, t, context, target)
            g = self._capture(args, key)
        if self.step_count + 1 - g["base"] >= self.GRAPH_WINDOW or g["lr"] != self.lr:
            self._graph_window(g)
        for dst, src in zip(g["inputs"], args):
            if dst is not None:
                dst.copy_(src, non_blocking=True)
        g["graph"].replay()
        # the host's mirror of what the replay did on the device
        self.step_count += 1
        if self.ema_decay is not None:
            self.ema_updates += 1
        ops.note_raw_write()                 # (the replayed optimizer wrote the parameters through raw pointers, like the eager launches it stands for)
        self._after_optimizer(g["fu"])
        g["token"], g["host_step"] = self.m.freshness_token(), self.step_count
        return g["loss"]

    def _capture(self, args, key):
        dev = args[0].device
        inputs = [None if a is None else torch.empty_like(a) for a in args]
        for dst, src in zip(inputs, args):
            if dst is not None:
                dst.copy_(src)
        g = {"key": key, "inputs": inputs, "sched": torch.empty((self.GRAPH_WINDOW, 4), dtype=torch.float32, device=dev),
             "idx": torch.zeros((1,), dtype=torch.int32, device=dev)}
        self._graph_window(g)
        # the capture pass runs the host side of one step without running its kernels: counters and flags are put back afterwards
        keep = (self.step_count, self.ema_updates, self.overlap_opt_fires, self.overlap_fires)
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        self._cap_sched = (g["sched"], g["idx"])
        try:
            with torch.cuda.graph(graph):
                ops.step_advance(g["idx"], 1)
                loss = self.train_step(*inputs)
        finally:
            self._cap_sched = None
            self.step_count, self.ema_updates, self.overlap_opt_fires, self.overlap_fires = keep
        fu = self._fused_opt() if self.fuse_packs else None
        g.update(graph=graph, loss=loss, fu=fu, token=self.m.freshness_token(), host_step=self.step_count)
        self._graph = g
        return g
