#!/usr/bin/env python3
"""Headline benchmark: U-Net denoising steps/sec (32x32x4 latent, bs=64 per GPU) on MI355X.

One "step" = one iteration of the reference's DDIM loop (ddim.py:139-160) for a batch of 64 latents with
classifier-free guidance: both U-Net evaluations (cond + uncond, ddim.py:177-178) + the fused CFG-rescale /
DDIM update. Synthetic inputs resident in HBM, PRNG-recipe weights of the NS32 architecture.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision f16|bf16|parity] [--batch 64]

`value` / `dtype` are the fastest numerics mode that meets north_star's tolerance (1e-3 relative to the fp32 reference, as rel-L2) on the
MEASURED deviation, single forward and over the whole DDIM-50 loop: fp16 single-product operands with fp32 accumulation (the reference is
fp32, train_diff.py:48). The bf16 single-product figure (BASELINE config 2's dtype; 6e-3 off the oracle) is a side field (`bf16_mode`).

N > 1: one rank per GPU; every rank denoises its own 64 latents (weak scaling, no collective inside the loop —
samples are independent); the final latents are all-gathered over RCCL once after the timed region. Launched either by
torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or by this script itself:
`python bench.py --gpus N` without WORLD_SIZE starts N rank processes (the parent never touches the GPU) and exits
with their return code. `--gpus` must equal WORLD_SIZE when both are given. Prints ONE JSON line on rank 0.

Rehearsal knobs for boxes without N GPUs (never set by the driver): STEDM_BENCH_ONE_DEVICE=1 puts every rank on
cuda:0, STEDM_BENCH_BACKEND=gloo replaces RCCL, STEDM_BENCH_DRY=1 skips the HIP work altogether (CPU tensors: only the
launcher, the process group, the barriers, the max-over-ranks timing and the all-gather run; the line says dry_run).
"""
from __future__ import annotations

from types import SimpleNamespace
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

NS32 = dict(image_size=32, in_channels=7, model_channels=128, out_channels=4, num_res_blocks=2,
            attention_resolutions=[32, 16, 8], channel_mult=[1, 4, 8], num_heads=8)
# algorithmic work: 54.288 GFLOP per U-Net sample-forward at 32x32 (hooks on the reference module)
GFLOP_PER_SAMPLE_FORWARD = 54.288
PEAK_MFMA_TFLOPS = 2500.0     # dense bf16/f16, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0


def build_model(dev, precision, use_graph=True):
    from stedm_amd.latent_diffusion import LatentDiffusion
    from stedm_amd.unet import UNetModel
    from stedm_amd.utils import prng
    unet = UNetModel(precision=precision, **NS32).eval()
    prng.fill_module_(unet, seed=0)
    ld = LatentDiffusion(unet, linear_start=0.0015, linear_end=0.0205, image_size=32, channels=4,
                         conditioning_key="hybrid", loss_type="l1", use_graph=use_graph)
    return ld.to(dev)


def synth_inputs(dev, B, rank, world=1):
    from stedm_amd import parallel as par
    from stedm_amd.utils import prng
    lo, hi = par.shard_range(B * world, rank, world)            # this rank's slice of the global batch
    ids = list(range(lo, hi))
    xT = par.per_sample_normal(1, ids, (4, 32, 32)).to(dev)       # per-sample streams: identical samples for any N
    layout = (par.per_sample_normal(2, ids, (3, 32, 32)) > 0).float().to(dev)   # stand-in for the rescaled layout
    ctx = par.per_sample_normal(3, ids, (512,)).to(dev)
    ctx_u = prng.normal(4, "bench.ctx_u", (1, 512)).repeat(B, 1).contiguous().to(dev)     # uncond style: one constant vector
    cond = {"c_concat": [layout], "c_crossattn": [ctx]}
    unc = {"c_concat": [layout], "c_crossattn": [ctx_u]}
    return xT, cond, unc


class ConvTimer:
    """Brackets every stedm_conv_igemm launch with HIP events on the launch stream (torch's current stream is
    the stream ops.py launches on) and records its algorithmic FLOPs = 2*M*N*K."""

    def __init__(self, split_gn: bool = False):
        """split_gn False: every launch is timed as the product issues it (a consumer GroupNorm riding on the convolution's epilogue /
        split-K reduce / trailing pass is inside the bracket: the shipped variant). True: that GroupNorm runs as its own launch after
        the bracket (convolution work only; the non-fused form)."""
        self.rec = []
        self.split_gn = split_gn

    def install(self):
        from stedm_amd import ops
        self._orig = ops.conv_igemm
        timer = self

        def timed(src1, w_hi, w_lo, out, **kw):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            gn_next = kw.pop("gn_next", None) if timer.split_gn else None
            e0.record()
            r = timer._orig(src1, w_hi, w_lo, out, **kw)
            e1.record()
            if gn_next is not None:
                g_w, g_b, g_eps, g_groups, g_act, g_out = gn_next[:6]
                g_lo = None
                if isinstance(g_out, (tuple, list)):          # 3-product modes: (hi, lo) planes
                    g_out, g_lo = g_out
                ops.gn_apply16c(out, kw["chan_stats"], None, None, g_out, g_lo, kw["prec"], g_w, g_b, g_eps, g_groups, g_act,
                                mean_rstd=gn_next[6] if len(gn_next) > 6 else None)
            if out is None:    # 16-bit-plane output only (qkv of the attention block; a decoder tensor written into the next concat's raw plane)
                out = kw["out16"][0]
            M = out.numel() // out.shape[-1]
            if kw.get("cout"):  # (out16_stride form: the plane is wider than the convolution's output)
                out = out[..., :kw["cout"]]
            if w_hi is None:   # space-to-depth Downsample: 9 taps x cin of the stride-2 conv (the 2x2 x 4cin form executes 16/9 of that)
                flops = 2.0 * M * out.shape[-1] * 9 * kw["src16"][0].shape[-1] / 4
            elif isinstance(w_hi, ops.LazyPlanes):   # fragment-order weights in use: K = taps x cin from the call (sub-pixel upsample: 4 executed taps)
                taps = 4 if kw.get("mode", 0) == ops.CONV_UP_SUBPIXEL else kw.get("ks", 3) ** 2
                flops = 2.0 * M * out.shape[-1] * taps * kw["src16"][0].shape[-1]
            else:
                flops = 2.0 * M * out.shape[-1] * w_hi.shape[1] * w_hi.shape[2]     # executed MACs (sub-pixel upsample: 4 taps, not 9)
            if kw.get("skip") is not None:                                       # fused 1x1 skip_connection phase
                flops += 2.0 * M * out.shape[-1] * kw["skip"][0].shape[-1]
            timer.rec.append((e0, e1, flops))
            return r

        ops.conv_igemm = timed

    def remove(self):
        from stedm_amd import ops
        ops.conv_igemm = self._orig

    def summary(self):
        torch.cuda.synchronize()
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in self.rec)
        fl = sum(f for _, _, f in self.rec)
        return {"launches": len(self.rec), "total_ms": ms, "avg_us": 1e3 * ms / max(1, len(self.rec)),
                "flops_per_launch": fl / max(1, len(self.rec)), "tflops": fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0}


class AttnTimer:
    """HIP-event bracket around every stedm_attn_legacy16 launch (the U-Net's MFMA attention): 4 * B * T^2 * C FLOP per call (QK^T and PV)."""

    def __init__(self):
        self.rec = []

    def install(self):
        from stedm_amd import ops
        self._orig = ops.attn_legacy16
        timer = self

        def timed(qkv, out16, heads, prec):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            r = timer._orig(qkv, out16, heads, prec)
            e1.record()
            B, T, C3 = qkv.shape
            timer.rec.append((e0, e1, 4.0 * B * T * T * (C3 // 3), T))
            return r

        ops.attn_legacy16 = timed

    def remove(self):
        from stedm_amd import ops
        ops.attn_legacy16 = self._orig

    def summary(self):
        torch.cuda.synchronize()
        ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in self.rec)
        fl = sum(f for _, _, f, _ in self.rec)
        return {"launches": len(self.rec), "total_ms": ms, "tflops": fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                "tokens": sorted({t for _, _, _, t in self.rec})}


REF128 = dict(image_size=128, in_channels=6, model_channels=128, out_channels=3, num_res_blocks=2,
              attention_resolutions=[32, 16, 8], channel_mult=[1, 4, 8], num_heads=8)
GFLOP_PER_SAMPLE_FORWARD_REF128 = 872.37      # BASELINE.md §3 (hooks on the reference module, 128x128x3 latents)


def ref128_leg(dev, precision, B, steps=6, warmup=2, spatial_transformer=False):
    """spatial_transformer: the same U-Net with use_spatial_transformer=True, context_dim=1024 (openaimodel.py:486, 644-667: the middle block's
    AttentionBlock becomes a SpatialTransformer, ldm/modules/attention.py:218-261, whose two CrossAttentions per block run as self-attentions
    at T = 1024, 8 heads x 128) - the north_star's "SpatialTransformer attention on MFMA tiles" leg.
    The reference-native shape (conf/diffusion/unet_config/landscape.yaml:1-16, conf/diffusion/ldm_based.yaml:10-11: 128x128x3 latents,
    Cin 6 / Cout 3; the only attention is the middle block's, T = 32 * 32 = 1024 tokens, openaimodel.py:644-649): the same DDIM-50 + CFG 1.5
    denoising step, hipGraph replay, at batch B per GPU; convolution roofline and the attention kernel's MFMA fraction from HIP events."""
    from stedm_amd.ddim import DDIMSampler, StepGraph
    from stedm_amd.latent_diffusion import LatentDiffusion
    from stedm_amd.unet import UNetModel
    from stedm_amd.utils import prng
    cfg = dict(REF128, use_spatial_transformer=True, context_dim=1024) if spatial_transformer else REF128
    unet = UNetModel(precision=precision, **cfg).eval()
    prng.fill_module_(unet, seed=0)
    ld = LatentDiffusion(unet, linear_start=0.0015, linear_end=0.0205, image_size=128, channels=3, conditioning_key="hybrid", loss_type="l1",
                         use_graph=True).to(dev)
    g = torch.Generator(device="cpu").manual_seed(13)
    xT = torch.randn(B, 3, 128, 128, generator=g).to(dev)
    lay = (torch.randn(B, 3, 128, 128, generator=g) > 0).float().to(dev)
    ctx, ctx_u = torch.randn(B, 512, generator=g).to(dev), torch.randn(1, 512, generator=g).repeat(B, 1).contiguous().to(dev)
    cond, unc = {"c_concat": [lay], "c_crossattn": [ctx]}, {"c_concat": [lay], "c_crossattn": [ctx_u]}
    dt, final = run_steps(ld, xT, cond, unc, warmup, steps, 1)
    assert bool(torch.isfinite(final).all())
    ms = 1e3 * dt / steps
    smp = DDIMSampler(ld); smp.make_schedule(50, ddim_eta=0.0, verbose=False)
    sg = StepGraph(smp, xT.clone(), cond, unc, 1.5)
    sg.reset(49); sg.step_eager(); torch.cuda.synchronize()
    ct, at = ConvTimer(), AttnTimer()
    ct.install(); at.install()
    for _ in range(2):
        sg.step_eager()
    cs, asum = ct.summary(), at.summary()
    ct.remove(); at.remove()
    fl = 2 * B * GFLOP_PER_SAMPLE_FORWARD_REF128 / 1e3
    rec = {"value": round(steps / dt, 3), "unit": "steps/s", "ms_per_step": round(ms, 3), "latent": "128x128x3", "batch": B, "dtype": unet.precision.label,
           "steps": steps, "warmup": warmup, "sample_steps_per_s": round(B * steps / dt, 1),
           "step_reference_equivalent_tflops": round(fl / (ms * 1e-3), 1),
           "step_reference_equivalent_frac_of_mfma_peak": round(fl / (ms * 1e-3) / PEAK_MFMA_TFLOPS, 4),
           "roofline": {"bound": "mfma", "kernel": "all stedm_conv_igemm launches of a REF128 step (as issued)", "achieved": round(cs["tflops"], 2),
                        "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(cs["tflops"] / PEAK_MFMA_TFLOPS, 4),
                        "launches_per_step": cs["launches"] // 2, "conv_ms_per_step": round(cs["total_ms"] / 2, 3)},
           "attention": {"kernel": "attn_flash_kernel (" + ("SpatialTransformer attn1 + attn2, self-attention" if spatial_transformer else "middle AttentionBlock") +
                                   "; workgroup = 128 queries of a sample-head, K / V rows of 64-key tiles staged once in LDS, online softmax)", "tokens": asum["tokens"],
                         "launches_per_step": asum["launches"] // 2, "us_per_launch": round(1e3 * asum["total_ms"] / max(1, asum["launches"]), 1),
                         "achieved": round(asum["tflops"], 2), "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(asum["tflops"] / PEAK_MFMA_TFLOPS, 4)},
           "what": "the reference-native U-Net (landscape.yaml: in 6 / out 3, 128^2 latents, 234.6 M parameters), one DDIM-50 + CFG 1.5 denoising step "
                   "per batch, hipGraph replay; 872.37 GFLOP per sample-forward"}
    if spatial_transformer:
        rec["what"] = ("the reference-native U-Net with use_spatial_transformer=True, context_dim=1024 (middle block: ResBlock, ResBlockStyle, "
                       "SpatialTransformer depth 1, ResBlock), one DDIM-50 + CFG 1.5 denoising step per batch, hipGraph replay")
        del rec["step_reference_equivalent_tflops"], rec["step_reference_equivalent_frac_of_mfma_peak"]     # (the 872.37 GFLOP count is the AttentionBlock net's)
    del sg, smp, ld, unet
    torch.cuda.empty_cache()
    return rec


def run_steps(ld, xT, cond, unc, warmup, steps, world):
    """W untimed + K timed denoising steps (hipGraph replay). Returns (seconds for K steps, final latents)."""
    from stedm_amd.ddim import DDIMSampler, StepGraph
    smp = DDIMSampler(ld, use_graph=True)
    smp.make_schedule(50, ddim_eta=0.0, verbose=False)      # BASELINE config 3: DDIM-50, eta 0, cfg 1.5
    n = smp.ddim_timesteps.shape[0]
    img = xT.clone()
    sg = StepGraph(smp, img, cond, unc, 1.5)
    sg.reset(n - 1)
    sg.step_eager()
    with sg.stream_ctx():
        sg.capture()
        left = n - 2
        for _ in range(max(0, warmup - 1)):
            if left < 0:
                sg.reset(n - 1); left = n - 1
            sg.replay(); left -= 1
        torch.cuda.current_stream().synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            if left < 0:
                sg.reset(n - 1); left = n - 1
            sg.replay(); left -= 1
        torch.cuda.synchronize()
        run_steps.own_seconds = time.perf_counter() - t0          # this rank's K steps alone (before the closing barrier)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    sg.join()
    return dt, img


def cpu_baseline(gpu_eval=None, gpu_loop=None):
    """The CPU oracle (fixture-pinned restatement of the reference's PyTorch-CPU path) on this host's cores: the WHOLE DDIM-50 + CFG 1.5 loop
    of the headline workload (50 denoising steps = 100 sequential U-Net forwards, as the reference does) on a bounded batch of 4 samples.
    gpu_eval (optional): callable(x, c_concat, ctx, ctx_u, t) -> {mode: (e_c, e_u)} evaluating the SAME 4 samples inside a bench-sized
    batch on the HIP path; the oracle's outputs then label every mode with its measured single-forward deviation (checker role).
    gpu_loop (optional): callable(x, c_concat, ctx, ctx_u) -> {mode: final latents of those 4 samples} after the same 50-step loop inside a
    bench-sized batch on the HIP path (hipGraph replay): the deviation every mode ACCUMULATES over the loop."""
    from oracle import ddim as od
    from oracle import unet as ou
    from stedm_amd.utils import prng
    torch.set_grad_enabled(False)
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # a 1-GPU box share is 16 host cores (the affinity mask shows the whole 256-thread host; using all of them
    # oversubscribes the share and runs ~50x slower). STEDM_CPU_THREADS overrides.
    cores = int(os.environ.get("STEDM_CPU_THREADS", min(cores, 16)))
    torch.set_num_threads(cores)
    cfg = ou.UNetConfig()
    plan = ou.build_plan(cfg)
    P = prng.fill_state_dict(plan.shapes, 0)
    Bc = 4
    x = prng.normal(1, "cpu.x", (Bc, 4, 32, 32)); cc = (prng.normal(2, "cpu.cc", (Bc, 3, 32, 32)) > 0).float()
    ctx = prng.normal(3, "cpu.ctx", (Bc, 512)); ctx_u = prng.normal(4, "cpu.ctxu", (Bc, 512))
    t = torch.full((Bc,), 951, dtype=torch.long)

    def apply_model(xx, tt, c):
        return ou.unet_forward(P, cfg, torch.cat([xx, c["c_concat"][0]], 1), tt, c["c_crossattn"][0], plan=plan)

    cond, unc = {"c_concat": [cc], "c_crossattn": [ctx]}, {"c_concat": [cc], "c_crossattn": [ctx_u]}

    def stat(got, ref):
        got, ref = got.double(), ref.double()
        return {"rel_l2": float((got - ref).norm() / ref.norm()), "max_over_std": float((got - ref).abs().max() / ref.std())}

    deviation = None
    if gpu_eval is not None:
        ref = torch.cat([apply_model(x, t, cond), apply_model(x, t, unc)])      # (also the warm-up of the timed loop below)
        deviation = {"rows": Bc, "what": "eps of 4 samples inside the bench-sized CFG batch (cond + uncond) vs the fp32 CPU oracle on the same samples: "
                                         "rel-L2 and max|diff|/std; north_star tolerance 1e-3"}
        for mode, (g_c, g_u) in gpu_eval(x, cc, ctx, ctx_u, 951).items():
            deviation[mode] = stat(torch.cat([g_c, g_u]), ref)
    else:
        apply_model(x, t, cond)
    S = int(os.environ.get("STEDM_CPU_LOOP_STEPS", "50"))
    t0 = time.perf_counter()
    final = od.ddim_sample(apply_model, od.Schedule(), x, cond, S, 0.0, uncond=unc, scale=1.5)
    el = time.perf_counter() - t0
    sample_steps_per_s = S * Bc / el
    out = {"value": sample_steps_per_s / 64.0, "unit": "steps/s (bs=64 equivalent)", "cores": cores, "kind": "port",
           "sample": f"the whole DDIM-{S} + CFG 1.5 loop ({S} denoising steps, {2 * S} U-Net forwards) at batch {Bc} (fp32 torch-CPU oracle, "
                     f"{cores} threads), {el:.1f} s; scaled by {Bc}/64", "sample_steps_per_s": sample_steps_per_s}
    loop_dev = None
    if gpu_loop is not None:
        loop_dev = {"rows": Bc, "steps": S,
                    "what": f"final latents of 4 samples after the whole DDIM-{S} + CFG 1.5 loop inside the bench-sized batch (hipGraph replay) vs the "
                            "fp32 CPU oracle's loop on the same samples: the deviation a mode accumulates over the loop; tolerance 1e-3 (rel-L2)"}
        for mode, got in gpu_loop(x, cc, ctx, ctx_u, S).items():
            loop_dev[mode] = stat(got, final)
    if cores != 8:
        # the second run BASELINE.md §4 promises: the same arithmetic on 8 threads, comparable with the 8-vCPU figures of BASELINE.md §2
        torch.set_num_threads(8)
        ds = od.DDIMSchedule(od.Schedule(), 50, 0.0)
        x8 = x
        n8, t0 = 0, time.perf_counter()
        while True:
            e8 = od.cfg_combine(apply_model(x8, t, cond), apply_model(x8, t, unc), 1.5)
            x8 = od.ddim_update(x8, e8, *ds.scalars(49))[0]; n8 += 1
            el8 = time.perf_counter() - t0
            if el8 > 4.0 or n8 >= 50:
                break
        out["threads8"] = {"value": n8 * Bc / el8 / 64.0, "unit": "steps/s (bs=64 equivalent)", "cores": 8,
                           "sample": f"{n8} CFG denoising steps at batch {Bc}, 8 threads, {el8:.1f} s; scaled by {Bc}/64",
                           "sample_steps_per_s": n8 * Bc / el8}
        torch.set_num_threads(cores)
    return out, deviation, loop_dev


def config1_leg(ld, dev, precision):
    """BASELINE config 1 (the reference's own CPU-runnable case): NS32, batch 1, DDIM-20 + CFG 1.5 — the whole loop on the HIP path
    (hipGraph replay through sample_log) beside the CPU oracle's loop on this host's cores, and the deviation of the final latent."""
    from oracle import ddim as od
    from oracle import unet as ou
    from stedm_amd.utils import prng
    cores = int(os.environ.get("STEDM_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16)))
    torch.set_num_threads(cores)
    cfg = ou.UNetConfig()
    plan = ou.build_plan(cfg)
    P = prng.fill_state_dict(plan.shapes, 0)
    xT = prng.normal(1, "c1.xT", (1, 4, 32, 32))
    cc = (prng.normal(2, "c1.layout", (1, 3, 32, 32)) > 0).float()
    ctx, ctx_u = prng.normal(3, "c1.ctx", (1, 512)), prng.normal(4, "c1.ctxu", (1, 512))

    def apply_model(x, t, c):
        return ou.unet_forward(P, cfg, torch.cat([x, c["c_concat"][0]], 1), t, c["c_crossattn"][0], plan=plan)

    t0 = time.perf_counter()
    ref = od.ddim_sample(apply_model, od.Schedule(), xT, {"c_concat": [cc], "c_crossattn": [ctx]}, 20, 0.0,
                         uncond={"c_concat": [cc], "c_crossattn": [ctx_u]}, scale=1.5)
    cpu_s = time.perf_counter() - t0
    unet = ld.model.diffusion_model
    cond = {"c_concat": [cc.to(dev)], "c_crossattn": [ctx.to(dev)]}
    unc = {"c_concat": [cc.to(dev)], "c_crossattn": [ctx_u.to(dev)]}
    res = {}
    for mode in dict.fromkeys([precision, "f16", "bf16", "parity"]):
        unet.set_precision(mode)
        run = lambda: ld.sample_log(cond, 1, True, 20, eta=0.0, x_T=xT.to(dev), unconditional_conditioning=unc, unconditional_guidance_scale=1.5,
                                    log_every_t=1000)[0]
        run()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        s = run()
        torch.cuda.synchronize()
        res[unet.precision.label] = {"seconds": round(time.perf_counter() - t1, 5),
                                     "rel_l2_vs_cpu_oracle": float((s.double().cpu() - ref.double()).norm() / ref.double().norm())}
    unet.set_precision(precision)
    return {"what": "NS32, batch 1, DDIM-20 + CFG 1.5 (rescale 0.7): the whole sampling loop, final latent compared with the CPU oracle's loop",
            "gpu": res, "cpu_oracle_seconds": round(cpu_s, 2), "cpu_cores": cores}


def cpu_baseline_train():
    """The CPU oracle's training-step arithmetic (forward + L1 + reverse-mode gradients by autograd over the fixture-pinned
    restatement; no optimizer) on this host's cores, one bounded sample."""
    from oracle import train as otrain
    from oracle import unet as ou
    from stedm_amd.utils import prng
    cores = int(os.environ.get("STEDM_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16)))
    torch.set_num_threads(cores)
    cfg = ou.UNetConfig()
    P = prng.fill_state_dict(ou.build_plan(cfg).shapes, 0)
    Bc = 2
    x = prng.normal(1, "cput.x", (Bc, 7, 32, 32)); ctx = prng.normal(3, "cput.ctx", (Bc, 512)); tgt = prng.normal(4, "cput.t", (Bc, 4, 32, 32))
    t = torch.tensor([951, 21], dtype=torch.long)
    t0 = time.perf_counter()
    otrain.unet_loss_and_grads(P, cfg, x, t, ctx, tgt)
    el = time.perf_counter() - t0
    torch.set_grad_enabled(False)
    return {"value": Bc / el, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"1 forward + L1 + backward at batch {Bc} (fp32 torch-CPU oracle under autograd, {cores} threads), {el:.1f} s"}


def train_leg_multi(ld, dev, args, xT, cond, rank, world, steps=4, warmup=2):
    """One data-parallel training step per rank at batch `args.batch` (weak scaling): returns the rank-0 record (max-over-ranks time)."""
    import torch.distributed as dist
    from stedm_amd.train import UNetTrainer
    B = args.batch
    unet = ld.model.diffusion_model
    unet.set_precision("bf16")               # BASELINE config 2's dtype
    tr = UNetTrainer(unet, lr=1e-6)
    g = torch.Generator(device="cpu").manual_seed(5 + rank)                 # every rank its own micro-batch
    tt = torch.randint(0, 1000, (B,), generator=g).to(dev)
    tgt = torch.randn(B, 4, 32, 32, generator=g).to(dev)
    xs, ccs, ctxs = xT, cond["c_concat"][0], cond["c_crossattn"][0]
    for _ in range(warmup):
        tr.train_step(xs, ccs, tt, ctxs, tgt)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.train_step(xs, ccs, tt, ctxs, tgt)
    torch.cuda.synchronize()
    own = (time.perf_counter() - t0) / steps                                 # this rank alone (its collectives wait for the slowest rank)
    dist.barrier()
    torch.cuda.synchronize()
    dtt = (time.perf_counter() - t0) / steps
    t = torch.tensor([dtt], dtype=torch.float64, device=dev)
    every = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(every, torch.tensor([own], dtype=torch.float64, device=dev))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # all ranks hold the same weights after the averaged step (the invariant DDP maintains): compare a checksum
    w = unet.out[2].weight.detach().float().sum().reshape(1).double()
    ws = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(ws, w)
    same = all(float(x.item()) == float(ws[0].item()) for x in ws)
    dmax = float(t.item())
    rec = {"ms": round(dmax * 1e3, 2), "steps_per_s": round(1 / dmax, 2), "samples_per_s": round(world * B / dmax, 1), "n_gpus": world,
           "batch_per_gpu": B, "global_batch": B * world, "ms_by_rank": [round(1e3 * float(x.item()), 2) for x in every],
           "overlapped_all_reduces": tr.overlap_fires, "buckets": len(tr._sched.bounds), "weights_equal_across_ranks": same,
           "dtype": unet.precision.label + " forward, bf16 backward operands, fp32 master/optimizer",
           "what": "forward + L1 + backward with bucketed gradient all-reduce (SUM, 1/N in the optimizer) started from inside the backward, fused "
                   "AdamW + EMA; weak scaling: every rank its own batch", "loss_rank0": round(float(loss), 4)}
    del tr
    return rec


def csrc_fingerprint():
    """sha1 over the sources a stedm_conv_igemm launch is compiled from (conv_rs.inc, conv_igemm*.{hip,inc}, conv_dma_*.hip,
    conv_common.hpp, common.hpp — not conv_io.hip, whose first/last-conv kernels are other entry points): a profile-derived figure
    (roofline.traffic) is only reported for the kernels it was taken on"""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "stedm_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".inc", ".hpp")) and ((f.startswith("conv") and f != "conv_io.hip") or f == "common.hpp"):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` outside a launcher: start N rank processes of this script (what torch.distributed.run does for
    train_diff.py:75 / predict_diff.py:86 in the reference) and return their exit code. Runs before anything initialises HIP."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0 and rc == 0:
                rc = r
                for q in live:          # a dead rank leaves the others waiting at a barrier: stop exactly the processes started here
                    q.terminate()
        time.sleep(0.05)
    return rc


def dry_run(args, rank, world):
    """STEDM_BENCH_DRY: the multi-rank plumbing of this file on CPU tensors (gloo), no HIP work, no throughput claim."""
    from stedm_amd import parallel as par
    if world > 1:
        torch.distributed.init_process_group("gloo")
    gb = int(os.environ.get("STEDM_BENCH_DRY_GLOBAL", args.batch * world))      # (a global batch the ranks do not divide: uneven shards)
    lo, hi = par.shard_range(gb, rank, world)
    final = par.per_sample_normal(1, list(range(lo, hi)), (4, 32, 32))
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        final = final * 1.0
    if world > 1:
        torch.distributed.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    ranks_seen = 1
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        gathered = par.all_gather_samples(final, gb)
        assert gathered.shape[0] == gb
        ranks_seen = torch.distributed.get_world_size()
        ref = par.per_sample_normal(1, list(range(gb)), (4, 32, 32))
        assert torch.equal(gathered, ref), "gathered samples differ from the single-rank stream"
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "U-Net denoising steps/sec (32x32x4 latent, bs=64)", "value": None, "unit": "steps/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "dry_run": True, "rccl_ranks": ranks_seen, "scaling": "weak",
                          "cpu_baseline": None if world == 1 else "n/a (world > 1): the CPU oracle is timed on rank 0 of the N = 1 run only",
                          "config": {"workload": "launcher rehearsal only (no HIP work)", "batch_per_gpu": args.batch,
                                     "global_batch": gb, "shard_sizes": [par.shard_range(gb, r, world)[1] - par.shard_range(gb, r, world)[0] for r in range(world)]}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--precision", default=os.environ.get("STEDM_BENCH_PRECISION", "f16"),
                    help="numerics mode of the headline: f16 (default: the fastest mode inside north_star's 1e-3), bf16, parity")
    ap.add_argument("--ref128-batch", type=int, default=16, help="batch of the reference-native 128x128x3 leg (0: skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-leg", action="store_true")
    ap.add_argument("--no-train-leg", action="store_true")
    ap.add_argument("--no-e2e-leg", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with matching values "
                         f"(python bench.py --gpus N starts the ranks itself)")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ.get("STEDM_BENCH_DRY"):
        return dry_run(args, rank, world)
    # rehearsal knobs for a one-GPU box (the driver's runs never set them): all ranks on one device, gloo instead of RCCL
    if os.environ.get("STEDM_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        backend = os.environ.get("STEDM_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)
    torch.set_grad_enabled(False)

    B = args.batch
    ld = build_model(dev, args.precision)
    xT, cond, unc = synth_inputs(dev, B, rank, world)
    dt, final = run_steps(ld, xT, cond, unc, args.warmup, args.steps, world)
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    rank_ms = [round(1e3 * run_steps.own_seconds / args.steps, 3)]
    if world > 1:
        every = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(every, torch.tensor([run_steps.own_seconds], dtype=torch.float64, device=dev))   # each rank's own K steps (stragglers show here)
        rank_ms = [round(1e3 * float(x.item()) / args.steps, 3) for x in every]
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        from stedm_amd import parallel as par
        gathered = par.all_gather_samples(final, B * world)                # prediction-side RCCL all-gather of the samples
        assert gathered.shape[0] == B * world and bool(torch.isfinite(gathered).all())
    ranks_seen = torch.distributed.get_world_size() if world > 1 else 1
    dt = float(t.item())
    steps_per_s = world * args.steps / dt
    ms_per_step = 1e3 * dt / args.steps

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel (conv_igemm): live HIP-event timing of every launch of one eager step
        from stedm_amd.ddim import DDIMSampler, StepGraph
        smp = DDIMSampler(ld)
        smp.make_schedule(50, ddim_eta=0.0, verbose=False)
        sg = StepGraph(smp, xT.clone(), cond, unc, 1.5)
        sg.reset(49)
        sg.step_eager()
        torch.cuda.synchronize()
        ct = ConvTimer(); ct.install()
        for _ in range(2):
            sg.step_eager()
        cs = ct.summary(); ct.remove()
        ct2 = ConvTimer(split_gn=True); ct2.install()        # the same launches with the riding GroupNorms split off (convolution work only)
        for _ in range(2):
            sg.step_eager()
        cs2 = ct2.summary(); ct2.remove()
        # HBM-side bytes per launch from the PMC passes (tools/pmc_traffic.py); reported only when the profile was taken on THIS
        # binary (fingerprint of the kernel sources), null otherwise — never a stale constant
        traffic, traffic_note = None, "no PMC traffic profile for this build (tools/pmc_traffic.py)"
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("csrc_fingerprint") == csrc_fingerprint():
                    traffic, traffic_note = tj.get("conv_igemm_hbm_bytes_per_launch"), "profiles/traffic.json (same kernel sources)"
                else:
                    traffic_note = "profiles/traffic.json was taken on other kernel sources: not reported"
            except Exception:
                pass
        roofline = {"bound": "mfma", "kernel": "all stedm_conv_igemm launches of a step: conv_rs_kernel (3x3, 3x3 + fused 1x1 skip, sub-pixel upsample, space-to-depth "
                              "downsample, 1x1) incl. their conv_splitk_reduce passes", "achieved": round(cs["tflops"], 2),
                    "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(cs["tflops"] / PEAK_MFMA_TFLOPS, 4),
                    "traffic": traffic, "traffic_source": traffic_note, "launches_per_step": cs["launches"] // 2, "avg_launch_us": round(cs["avg_us"], 2),
                    "algorithmic_gflop_per_launch": round(cs["flops_per_launch"] / 1e9, 3),
                    "conv_ms_per_step": round(cs["total_ms"] / 2, 3), "mode": ld.model.diffusion_model.precision.label,
                    "timed_variant": "the launches as the step issues them: a consumer GroupNorm + SiLU that rides on a convolution (epilogue form, "
                                     "split-K reduce form, or the trailing stedm_gn_apply16c pass of the call) is INSIDE the bracket, its bytes are "
                                     "not counted as work",
                    "convolution_only": {"achieved": round(cs2["tflops"], 2), "frac": round(cs2["tflops"] / PEAK_MFMA_TFLOPS, 4),
                                         "conv_ms_per_step": round(cs2["total_ms"] / 2, 3),
                                         "what": "the same launches with every riding GroupNorm run as its own launch outside the bracket"}}
        # whole-step figures against the same peak: (a) reference-equivalent = the FLOPs of the reference's two full forwards per step
        # (what a user gets per second, in the reference's currency); (b) executed = the conv FLOPs the hardware really runs per step
        # (shared encoder evaluated once, sub-pixel upsample at 4/9 of the MACs): the honest MFMA utilisation of the whole step
        step_tflops = 2 * B * GFLOP_PER_SAMPLE_FORWARD / 1e3 / (ms_per_step * 1e-3) * 1.0
        exec_tflops = cs["flops_per_launch"] * (cs["launches"] // 2) / 1e12 / (ms_per_step * 1e-3)
        out = {
            "metric": "U-Net denoising steps/sec (32x32x4 latent, bs=64)", "value": round(steps_per_s, 3), "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": ld.model.diffusion_model.precision.label,
            "data": "synthetic",
            "config": {"workload": "HER2-style DDIM-50 + CFG 1.5 (rescale 0.7) denoising step, NS32 U-Net (234.6M params, "
                                   "random-init PRNG weights), 32x32x4 latents + 3-ch layout concat, style vector 512",
                       "batch_per_gpu": B, "global_batch": B * world, "latent": "32x32x4", "cfg": "cond+uncond per step",
                       "parallelism": f"dp{world} (independent latents, no in-loop collective)"},
            "sample_steps_per_s": round(steps_per_s * B, 1),
            "rccl_ranks": ranks_seen,
            "ms_per_step_by_rank": rank_ms,
            "step_reference_equivalent_tflops": round(step_tflops, 1),
            "step_reference_equivalent_frac_of_mfma_peak": round(step_tflops / PEAK_MFMA_TFLOPS, 4),
            "step_executed_conv_tflops": round(exec_tflops, 1), "step_executed_frac_of_mfma_peak": round(exec_tflops / PEAK_MFMA_TFLOPS, 4),
            "roofline": roofline,
        }
        if not args.no_parity_leg and world == 1:
            del sg, smp
            # the other numerics modes on the same workload, batch, graph, steps and warm-up as the headline
            unet_ = ld.model.diffusion_model
            for mode, leg, note in (("bf16", "bf16_mode", "bf16 single product (BASELINE config 2's dtype): outside the 1e-3 tolerance, see deviation_vs_cpu_oracle.bf16"),
                                    ("f16", "f16_mode", "fp16 single product; measured deviation from the CPU oracle: deviation_vs_cpu_oracle.f16"),
                                    ("parity", "parity_mode", "fp16 hi + lo split operands, 3 products: meets 1e-3 under the max-norm reading too "
                                                              "(deviation_vs_cpu_oracle.f16x3; tests/test_gpu_unet.py, tests/test_gpu_bench_config.py)")):
                if mode == args.precision:
                    continue
                unet_.set_precision(mode)
                dtm, _ = run_steps(ld, xT, cond, unc, args.warmup, args.steps, 1)
                out[leg] = {"dtype": unet_.precision.label, "value": round(args.steps / dtm, 3), "unit": "steps/s", "steps": args.steps,
                            "warmup": args.warmup, "note": note}
            unet_.set_precision(args.precision)
            if args.ref128_batch > 0:
                try:
                    out["ref128_step"] = ref128_leg(dev, args.precision, args.ref128_batch)
                except Exception as e:          # never lose the headline over a side leg
                    out["ref128_step"] = {"error": str(e)[:200]}
                try:
                    out["st_step"] = ref128_leg(dev, args.precision, args.ref128_batch, spatial_transformer=True)
                except Exception as e:
                    out["st_step"] = {"error": str(e)[:200]}
        if not args.no_e2e_leg and world == 1:
            # BASELINE config 5's latent size (64x64x4, CATCH 512^2): the same denoising step on 4x the pixels, reported beside the headline
            ld.model.diffusion_model.set_precision(args.precision)
            g = torch.Generator(device="cpu").manual_seed(11)
            x64 = torch.randn(B, 4, 64, 64, generator=g).to(dev)
            lay64 = (torch.randn(B, 3, 64, 64, generator=g) > 0).float().to(dev)
            c64 = {"c_concat": [lay64], "c_crossattn": cond["c_crossattn"]}
            u64 = {"c_concat": [lay64], "c_crossattn": unc["c_crossattn"]}
            dt64, _ = run_steps(ld, x64, c64, u64, 2, 6, 1)
            fl64 = 2 * B * 217.31 / 1e3        # TFLOP per CFG step (217.31 GFLOP per sample-forward at 64^2)
            out["ns64_step"] = {"value": round(6 / dt64, 3), "unit": "steps/s", "ms_per_step": round(1e3 * dt64 / 6, 3), "latent": "64x64x4", "batch": B,
                                "step_reference_equivalent_tflops": round(fl64 / (dt64 / 6), 1),
                                "step_reference_equivalent_frac_of_mfma_peak": round(fl64 / (dt64 / 6) / PEAK_MFMA_TFLOPS, 4)}
            del x64, lay64
            # BASELINE config 5's per-GPU slice of the style encoding: 8 style inputs per sample through the set encoder (patch features 192 * 8 wide)
            # with MX-fp8 attention operands, beside the bf16 attention; the 64x64x4 denoising step above is the same slice's U-Net half
            try:
                from stedm_amd.style import sViT
                from stedm_amd.utils import prng as _prng
                Bs = min(B, 32)              # (the 8-image style stack of 64 samples is 6.4 GB of fp32 pixels: half a batch is timed)
                sv = sViT(image_size=512, patch_size=8, num_classes=512, dim=256, depth=6, heads=12, mlp_dim=256, pool="mean", channels=3, dropout=0.1,
                          emb_dropout=0.1, ns=8, t_dim=256, precision=args.precision).eval()
                _prng.fill_module_(sv, seed=7)
                sv = sv.to(dev)
                sty8 = torch.rand(Bs, 8, 512, 512, 3, device=dev) * 2 - 1
                t8 = {}
                y8 = {}
                for mode in (args.precision, "fp8"):
                    sv.set_precision(mode)
                    sv(sty8)
                    torch.cuda.synchronize(); ts0 = time.perf_counter()
                    y8[mode] = sv(sty8).float().clone()
                    torch.cuda.synchronize(); t8[mode] = time.perf_counter() - ts0
                out["config5_style_slice"] = {"batch": Bs, "style_images_per_sample": 8, "seconds": {args.precision: round(t8[args.precision], 4), "mx_fp8_attention": round(t8["fp8"], 4)},
                                              "samples_per_s": {args.precision: round(Bs / t8[args.precision], 1), "mx_fp8_attention": round(Bs / t8["fp8"], 1)},
                                              "fp8_vs_" + args.precision + "_rel_l2": round(float((y8["fp8"] - y8[args.precision]).norm() / y8[args.precision].norm()), 5),
                                              "what": "sViT(ns=8) over 8 x 512^2 style images per sample (CATCH-style set aggregation), 4098 tokens, 6 layers x 12 heads"}
                del sv, sty8, y8
            except Exception as e:
                out["config5_style_slice"] = {"error": str(e)[:160]}
            # BASELINE config 3 end to end up to the sampled latents: style encoder (sViT, 4 style images of 512^2 per sample) + layout
            # rescaler + DDIM-50 with CFG, the call sequence of LDM_Diffusion.predict_step
            from stedm_amd.latent_diffusion import S_ZSS_DM, predict_latents
            from stedm_amd.utils import prng
            unet = ld.model.diffusion_model
            unet.set_precision(args.precision)
            agg = dict(name="svit", patch_size=8, dim=256, depth=6, heads=12, mlp_dim=256, pool="mean", channels=3, dropout=0.1, emb_dropout=0.1,
                       t_dim=256)
            # first stage: the vq-f4.yaml architecture (ch 128, ch_mult 1-2-4, 2 res blocks, 8192-entry codebook) with 4 latent channels
            # for the synthetic 32x32x4 latents -> 128x128x3 images
            fs_cfg = {"target": "ldm.models.autoencoder.VQModelInterface",
                      "params": dict(embed_dim=4, n_embed=8192, lossconfig={"target": "torch.nn.Identity"}, precision=args.precision,
                                     ddconfig=dict(double_z=False, z_channels=4, resolution=128, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4],
                                                   num_res_blocks=2, attn_resolutions=[], dropout=0.0))}
            zm = S_ZSS_DM("swin_v2_t", dict(name="mp", num_patches=4), agg, {"data": {"patch_size": 512}}, unet, linear_start=0.0015,
                          linear_end=0.0205, image_size=32, channels=4, conditioning_key="hybrid", loss_type="l1", cond_stage_key="segmentation",
                          use_graph=True, first_stage_config=fs_cfg,
                          cond_stage_config={"target": "ldm.modules.encoders.modules.SpatialRescaler",
                                             "params": {"n_stages": 3, "in_channels": 2, "out_channels": 3}})
            prng.fill_module_(zm.agg_block, seed=51)
            prng.fill_module_(zm.cond_stage_model, seed=52)
            prng.fill_module_(zm.first_stage_model, seed=53)
            zm = zm.to(dev).eval()
            zm.agg_block.set_precision(args.precision)
            g = torch.Generator(device="cpu").manual_seed(7)
            batch = {"image": torch.zeros(B, 256, 256, 3, device=dev),
                     "segmentation": (torch.rand(B, 256, 256, 2, generator=g) > 0.5).float().to(dev),
                     "style_imgs": (torch.rand(B, 4, 512, 512, 3, generator=g) * 2 - 1).to(dev)}
            from stedm_amd.latent_diffusion import images_for_saving

            def predict_step():          # LDM_Diffusion.predict_step, ldm_diffusion.py:76-99, up to the uint8 arrays
                lat = predict_latents(zm, batch, ddim_steps=50, cfg_scale=1.5, x_T=xT)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                img, seg = images_for_saving(zm.decode_first_stage(lat), batch["segmentation"])
                torch.cuda.synchronize()
                return lat, img, t1

            predict_step()          # warm-up: packs weights, captures the step graph
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lat, img, t1 = predict_step()
            t2 = time.perf_counter()
            dte = t1 - t0
            assert bool(torch.isfinite(lat).all()) and tuple(img.shape) == (B, 128, 128, 3)
            out["sampling_run"] = {"seconds": round(dte, 4), "latents_per_s": round(B / dte, 1), "batch": B, "ddim_steps": 50,
                                   "what": "S_ZSS_DM.get_input (sViT over 4 x 512^2 style images per sample + SpatialRescaler) + unconditional style vector "
                                           "(one constant sample, broadcast) + DDIM-50 with CFG 1.5: predict_step up to the sampled latents",
                                   "decode_seconds": round(t2 - t1, 4), "images_per_s": round(B / (t2 - t0), 1),
                                   "decode_what": "decode_first_stage (VQ-f4 architecture: quantise over 8192 codes, post_quant_conv, Decoder 32^2 -> 128^2, "
                                                  "55 M parameters) + uint8 / class-map epilogue: predict_step end to end, images/s over the whole step"}
            # PCIe-inclusive figure: predict_step of the reference receives its batch from the DataLoader on the HOST (ldm_diffusion.py:76-79);
            # the legs above start with the batch resident in HBM. Time the host -> device copy of the same batch from pinned memory
            # (what DataLoader(pin_memory=True) hands over) and report the end-to-end rate with it, beside the resident one — never as `value`.
            try:
                from stedm_amd.parallel import BatchPrefetcher
                hb = {k: v.cpu().pin_memory() for k, v in batch.items()}
                nbytes = sum(v.numel() * v.element_size() for v in hb.values())
                for _ in range(2):
                    torch.cuda.synchronize(); th0 = time.perf_counter()
                    db = {k: v.to(dev, non_blocking=True) for k, v in hb.items()}
                    torch.cuda.synchronize(); th1 = time.perf_counter()
                del db
                # pipelined: batch i + 1 crosses PCIe on a side stream while batch i samples (stedm_amd.parallel.BatchPrefetcher); the loop
                # below is predict_step over 3 host batches, every image counted, nothing resident in HBM beforehand but the first copy's start
                pf = BatchPrefetcher(dev)
                nb = 3
                torch.cuda.synchronize(); tp0 = time.perf_counter()
                h = pf.submit(hb)
                for i in range(nb):
                    cur = pf.get(h)
                    if i + 1 < nb:
                        h = pf.submit(hb)
                    lat = predict_latents(zm, cur, ddim_steps=50, cfg_scale=1.5, x_T=xT)
                    img, seg = images_for_saving(zm.decode_first_stage(lat), cur["segmentation"])
                    del cur
                torch.cuda.synchronize(); tp1 = time.perf_counter()
                out["sampling_run"]["h2d"] = {"seconds": round(th1 - th0, 4), "mbytes": round(nbytes / 1e6, 1), "gb_per_s": round(nbytes / 1e9 / (th1 - th0), 1),
                                              "images_per_s_incl_h2d_serial": round(B / (t2 - t0 + th1 - th0), 1),
                                              "images_per_s_incl_h2d": round(nb * B / (tp1 - tp0), 1),
                                              "what": "pinned host batch (image, segmentation, 4 style images of 512^2 per sample, fp32) -> HBM; _serial: the copy "
                                                      "in front of the step; images_per_s_incl_h2d: 3 host batches through predict_step with the next "
                                                      "batch's copy on a side stream beside the current batch's sampling (first copy exposed)"}
                del hb
            except RuntimeError as e:            # no pinned memory on this box: say so instead of a number
                out["sampling_run"]["h2d"] = {"error": str(e)[:120]}
            # BASELINE config 5's numerics mode for the style encoder: MX-fp8 attention operands (v_mfma_scale_f32_32x32x64_f8f6f4) beside the
            # bf16 attention, same sViT, same B x 4 x 512^2 style images; deviations of both modes: tests/test_gpu_style.py
            try:
                sty = batch["style_imgs"]
                tms = {}
                for mode in (args.precision, "fp8"):
                    zm.agg_block.set_precision(mode)
                    zm.agg_block(sty)
                    torch.cuda.synchronize(); ts0 = time.perf_counter()
                    zm.agg_block(sty)
                    torch.cuda.synchronize(); tms[mode] = time.perf_counter() - ts0
                zm.agg_block.set_precision(args.precision)
                out["sampling_run"]["svit_style_seconds"] = {args.precision: round(tms[args.precision], 4), "mx_fp8_attention": round(tms["fp8"], 4),
                                                             "what": f"sViT forward over {B} x 4 style images of 512^2 (4098 tokens, 6 layers x 12 heads): "
                                                                     "bf16 attention vs MX-fp8 attention operands (everything else bf16)"}
            except Exception as e:      # never lose the line over the side measurement
                out["sampling_run"]["svit_style_seconds"] = {"error": str(e)[:160]}
            # the default aggregator of the reference's config (conf/config_diff.yaml:16 style_agg: linear): Swin-V2-T over the same B x 4 style
            # images + the Agg_Linear MLP (networks/agg_blocks.py:24-33) — the style encoding that replaces the sViT's in predict_step
            zs = S_ZSS_DM("swin_v2_t", SimpleNamespace(name="mp", num_patches=4), SimpleNamespace(name="linear"), {"data": {"patch_size": 512}}, unet,
                          linear_start=0.0015, linear_end=0.0205, image_size=32, channels=4, conditioning_key="hybrid", loss_type="l1",
                          cond_stage_key="segmentation")
            prng.fill_module_(zs.agg_block.linear_block, seed=54)
            zs = zs.to(dev).eval()
            zs.agg_block.set_precision(args.precision)
            zs.agg_block(batch["style_imgs"])
            torch.cuda.synchronize(); t3 = time.perf_counter()
            sv = zs.agg_block(batch["style_imgs"])
            torch.cuda.synchronize(); t4 = time.perf_counter()
            assert tuple(sv.shape) == (B, 512) and bool(torch.isfinite(sv).all())
            out["sampling_run"]["swin_linear_style_seconds"] = round(t4 - t3, 4)
            out["sampling_run"]["swin_linear_what"] = (f"Agg_Linear over swin_v2_t (stedm_amd/swin.py, trunc-normal weights) on {B} x 4 style images of 512^2: "
                                                       f"{4 * B / (t4 - t3):.0f} images/s, 47.4 GFLOP per image; parity of the Swin restatement is UNPINNED "
                                                       f"(torchvision absent, the reference holds no fixture)")
            del zm, zs, batch
        if not args.no_train_leg and world == 1:
            # BASELINE config 2: one training step (forward + L1 + backward + AdamW/EMA) on the same U-Net and batch
            from stedm_amd.train import UNetTrainer
            unet = ld.model.diffusion_model
            unet.set_precision("bf16")           # BASELINE config 2 names bf16: forward and backward operands bf16 (fp32 exponent range, no loss scaling)
            tr = UNetTrainer(unet, lr=1e-6)
            g = torch.Generator(device="cpu").manual_seed(5)
            tt = torch.randint(0, 1000, (B,), generator=g).to(dev)
            tgt = torch.randn(B, 4, 32, 32, generator=g).to(dev)
            xs, ccs, ctxs = xT, cond["c_concat"][0], cond["c_crossattn"][0]
            for _ in range(2):
                tr.train_step(xs, ccs, tt, ctxs, tgt)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                loss = tr.train_step(xs, ccs, tt, ctxs, tgt)
            torch.cuda.synchronize()
            dte = (time.perf_counter() - t0) / 4
            # the same step captured once and replayed as one hipGraph launch (UNetTrainer.train_step_graphed: bitwise the eager step,
            # tests/test_gpu_train.py); the capture happens on the first call after GRAPH_WARMUP eager ones
            for _ in range(tr.GRAPH_WARMUP + 2):
                tr.train_step_graphed(xs, ccs, tt, ctxs, tgt)
            assert tr._graph is not None
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(6):
                loss = tr.train_step_graphed(xs, ccs, tt, ctxs, tgt)
            t_issue = (time.perf_counter() - t0) / 6
            torch.cuda.synchronize()
            dtt = (time.perf_counter() - t0) / 6
            out["train_step"] = {"ms": round(dtt * 1e3, 2), "steps_per_s": round(1 / dtt, 2), "samples_per_s": round(B / dtt, 1),
                                 "launch": "hipGraph replay (one launch per step)", "host_ms_per_step": round(t_issue * 1e3, 2),
                                 "eager_ms": round(dte * 1e3, 2),
                                 "dtype": unet.precision.label + " forward, bf16 backward operands, fp32 master/optimizer", "batch": B,
                                 "algorithmic_tflops": round(3 * B * GFLOP_PER_SAMPLE_FORWARD / 1e3 / dtt, 1),
                                 "what": "forward + L1 loss + backward (dgrad on the forward's MFMA conv kernels, direct 3x3 wgrad kernel) + fused AdamW + EMA, 234.6M params; "
                                         "gradients match the reference's autograd to 1e-5 in parity mode (tests/test_gpu_train.py)",
                                 "loss": round(float(loss), 4)}
            del tr
            unet.set_precision(args.precision)
            if not args.no_cpu_baseline:
                out["train_step"]["cpu_baseline"] = cpu_baseline_train()
        if not args.no_cpu_baseline and world == 1:
            unet = ld.model.diffusion_model

            def gpu_eval(x4, cc4, ctx4, ctxu4, tval):
                # the oracle's 4 samples as rows 0..3 of a bench-sized batch (rows are independent): the kernels selected are the bench's
                from stedm_amd.utils import prng
                prng.fill_module_(unet, seed=0)          # the training leg stepped the weights: back to the PRNG recipe the oracle holds
                unet.invalidate()
                n4 = x4.shape[0]
                xb, ccb = xT.clone(), cond["c_concat"][0].clone()
                cb, ub = cond["c_crossattn"][0].clone(), unc["c_crossattn"][0].clone()
                xb[:n4], ccb[:n4], cb[:n4], ub[:n4] = x4.to(dev), cc4.to(dev), ctx4.to(dev), ctxu4.to(dev)
                tb = torch.full((B,), int(tval), dtype=torch.long, device=dev)
                res = {}
                for mode in dict.fromkeys([args.precision, "f16", "bf16", "parity"]):
                    unet.set_precision(mode)
                    ec, eu = unet.forward_cfg(xb, ccb, tb, cb, ub, uniform_t=True)
                    res[unet.precision.label] = (ec[:n4].float().cpu(), eu[:n4].float().cpu())
                unet.set_precision(args.precision)
                return res

            def gpu_loop(x4, cc4, ctx4, ctxu4, S):
                # the same 4 samples through the WHOLE loop inside a bench-sized batch (hipGraph replay, as the headline runs it), every mode
                n4 = x4.shape[0]
                xb, ccb = xT.clone(), cond["c_concat"][0].clone()
                cb, ub = cond["c_crossattn"][0].clone(), unc["c_crossattn"][0].clone()
                xb[:n4], ccb[:n4], cb[:n4], ub[:n4] = x4.to(dev), cc4.to(dev), ctx4.to(dev), ctxu4.to(dev)
                c_, u_ = {"c_concat": [ccb], "c_crossattn": [cb]}, {"c_concat": [ccb], "c_crossattn": [ub]}
                res = {}
                for mode in dict.fromkeys([args.precision, "f16", "bf16", "parity"]):
                    unet.set_precision(mode)
                    fin = ld.sample_log(c_, B, True, S, eta=0.0, x_T=xb, unconditional_conditioning=u_, unconditional_guidance_scale=1.5,
                                        log_every_t=1000)[0]
                    res[unet.precision.label] = fin[:n4].float().cpu()
                unet.set_precision(args.precision)
                return res

            out["cpu_baseline"], dev_rep, loop_rep = cpu_baseline(gpu_eval=gpu_eval, gpu_loop=gpu_loop)
            out["config1_loop"] = config1_leg(ld, dev, args.precision)      # (after gpu_eval: it restores the PRNG weights the oracle holds)
            out["deviation_vs_cpu_oracle"] = dev_rep
            out["loop_deviation_vs_cpu_oracle"] = loop_rep
            if dev_rep and out["dtype"] in dev_rep:
                out["headline_rel_l2_vs_oracle"] = round(dev_rep[out["dtype"]]["rel_l2"], 6)
            if loop_rep and out["dtype"] in loop_rep:
                out["headline_loop_rel_l2_vs_oracle"] = round(loop_rep[out["dtype"]]["rel_l2"], 6)
            if dev_rep and loop_rep:
                out["headline_meets_tolerance"] = bool(out.get("headline_rel_l2_vs_oracle", 1.0) <= 1e-3 and
                                                       out.get("headline_loop_rel_l2_vs_oracle", 1.0) <= 1e-3)
                for leg in ("bf16_mode", "f16_mode", "parity_mode"):
                    if leg in out and out[leg]["dtype"] in dev_rep:
                        out[leg]["rel_l2_vs_oracle"] = round(dev_rep[out[leg]["dtype"]]["rel_l2"], 6)
                        out[leg]["loop_rel_l2_vs_oracle"] = round(loop_rep[out[leg]["dtype"]]["rel_l2"], 6)
                # the throughput of the fastest mode that meets north_star's 1e-3 (fp32-relative) on the measured deviation — single forward
                # AND accumulated over the DDIM-50 loop — first class; the headline is that mode unless --precision asked for another
                rates = {out["dtype"]: out["value"]}
                for leg in ("bf16_mode", "f16_mode", "parity_mode"):
                    if leg in out:
                        rates[out[leg]["dtype"]] = out[leg]["value"]

                def fastest(key):
                    ok = [(rates[m], m) for m in rates if m in dev_rep and m in loop_rep and dev_rep[m][key] <= 1e-3 and loop_rep[m][key] <= 1e-3]
                    if not ok:
                        return None
                    v, m = max(ok)
                    return {"mode": m, "value": v, "unit": "steps/s", "rel_l2": round(dev_rep[m]["rel_l2"], 8),
                            "max_over_std": round(dev_rep[m]["max_over_std"], 8), "loop_rel_l2": round(loop_rep[m]["rel_l2"], 8),
                            "loop_max_over_std": round(loop_rep[m]["max_over_std"], 8)}
                out["value_at_tolerance"] = {"tolerance": 1e-3, "on_rel_l2": fastest("rel_l2"), "on_max_over_std": fastest("max_over_std"),
                                             "what": "fastest measured mode whose eps (one forward) AND final latents (whole DDIM-50 loop) deviate from "
                                                     "the fp32 CPU oracle by <= 1e-3 — as rel-L2, and under the stricter max|diff|/std reading; same "
                                                     "workload, batch, graph, steps and warm-up as `value`"}
        elif world == 1:
            out["cpu_baseline"] = None
        else:
            out["cpu_baseline"] = "n/a (world > 1): the CPU oracle is timed on rank 0 of the N = 1 run only"
    if world > 1 and not args.no_train_leg:
        # the training half of north_star on N ranks (train_diff.py:75-76: DDP over the ranks, one process per GPU): BASELINE config 2's step
        # at the per-GPU batch — forward + L1 + backward with the gradient buckets all-reduced over RCCL while the backward still runs, then
        # fused AdamW + EMA on the rank-averaged gradients. Timed like the headline: barrier + synchronize on both sides, max over ranks.
        tl = train_leg_multi(ld, dev, args, xT, cond, rank, world)
        if rank == 0:
            out["train_step"] = tl
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
