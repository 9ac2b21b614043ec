"""GPU tier: the configurations BASELINE.json names, at THEIR sizes, against the CPU oracle directly (not against another mode of
the HIP path). Rows of a batch are independent (GroupNorm and the CFG std are per sample), so a few rows of a bench-sized batch are
checked against the oracle run on exactly those samples.

  config 3 / metric  NS32, batch 64, CFG pass (128 decoder rows): register-streamed 3x3 / fused-skip / split-K / sub-pixel / s2d kernels
  config 1           NS32, batch 1, DDIM-20 + CFG 1.5 through sample_log(use_graph=True): the 16-way split-K path through a whole loop
  config 2           NS32, batch 64, bf16 training step
north_star tolerance: 1e-3 relative fp32, asserted in the parity mode (fp16 x3) and, for rel-L2, in the fp16 single-product mode; the
bf16 single-product deviation (the dtype BASELINE names for the headline) is measured, printed and bounded loosely."""
import numpy as np
import pytest
import torch

from stedm_amd.utils import prng

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

NS32 = dict(image_size=32, in_channels=7, model_channels=128, out_channels=4, num_res_blocks=2,
            attention_resolutions=[32, 16, 8], channel_mult=[1, 4, 8], num_heads=8)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def oracle_ns32():
    from oracle import unet as ou
    cfg = ou.UNetConfig()
    plan = ou.build_plan(cfg)
    return cfg, plan, prng.fill_state_dict(plan.shapes, 0)


def build(dev, precision):
    from stedm_amd.unet import UNetModel
    m = UNetModel(precision=precision, **NS32).eval()
    prng.fill_module_(m, seed=0)
    return m.to(dev)


def dev2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm()), float((a - b).abs().max() / b.std())


def test_batch64_cfg_rows_vs_cpu_oracle(dev, oracle_ns32):
    """4 fixed rows (first, two inner, last) of the B=64 CFG pass, cond and uncond halves, against oracle.unet.unet_forward on those
    4 samples. parity: < 1e-3 (both measures); f16: rel-L2 < 1e-3; bf16: reported."""
    from oracle import unet as ou
    cfg, plan, P = oracle_ns32
    B = 64
    x = prng.normal(3, "bc.x", (B, 4, 32, 32)); cc = prng.normal(3, "bc.cc", (B, 3, 32, 32))
    ctx_c = prng.normal(3, "bc.ctx", (B, 512)); ctx_u = prng.normal(3, "bc.ctxu", (B, 512))
    rows = [0, 21, 42, 63]
    t4 = torch.full((4,), 951, dtype=torch.long)
    xc4 = torch.cat([x[rows], cc[rows]], 1)
    ref = torch.cat([ou.unet_forward(P, cfg, xc4, t4, ctx_c[rows], plan=plan), ou.unet_forward(P, cfg, xc4, t4, ctx_u[rows], plan=plan)])
    m = build(dev, "parity")
    t = torch.full((B,), 951, dtype=torch.long, device=dev)
    out = {}
    for precision in ("parity", "f16", "bf16"):
        m.set_precision(precision)
        ec, eu = m.forward_cfg(x.to(dev), cc.to(dev), t, ctx_c.to(dev), ctx_u.to(dev), uniform_t=True)
        out[precision] = dev2(torch.cat([ec[rows], eu[rows]]), ref)
        print(f"[NS32 B=64 CFG, rows {rows}, {precision}] vs CPU oracle: rel-L2 {out[precision][0]:.3e}, max/std {out[precision][1]:.3e}")
    assert out["parity"][0] < 1e-3 and out["parity"][1] < 1e-3
    # f16 (the bench headline's mode): inside north_star's 1e-3 as rel-L2 (measured 7.5e-4 .. 7.9e-4, round 5); the max-norm reading is
    # bounded at what is measured (4.0e-3 .. 4.5e-3) + margin, so a regression shows - the mode that meets 1e-3 on BOTH readings is `parity`
    assert out["f16"][0] < 1e-3 and out["f16"][1] < 5.5e-3
    assert out["bf16"][0] < 1.5e-2 and out["bf16"][1] < 8e-2


def test_batch256_cfg_rows_vs_cpu_oracle(dev, oracle_ns32):
    """The hazard that round 3's aliasing bug lived in: at B = 64 every NS32 launch has one tile round per CU (256 or 512 tiles on 256 CUs), so
    a tile that overwrites rows another tile still gathers is never seen. B = 256 (512 decoder rows: 1024 - 2048 tiles, >= 4 rounds per
    CU) drives the same kernels through several rounds: the epilogue-GroupNorm form (8^2 / 16^2 levels), the gn_only form (no fp32 store), the
    16-bit-only output in front of the Upsample, the split-K reduce with GroupNorm — rows from the first, inner and last tiles against the CPU
    oracle on those samples (openaimodel.py:247-254, 288; :122-132)."""
    from oracle import unet as ou
    cfg, plan, P = oracle_ns32
    B = 256
    x = prng.normal(8, "b256.x", (B, 4, 32, 32)); cc = prng.normal(8, "b256.cc", (B, 3, 32, 32))
    ctx_c = prng.normal(8, "b256.ctx", (B, 512)); ctx_u = prng.normal(8, "b256.ctxu", (B, 512))
    rows = [0, 101, 202, 255]
    t4 = torch.full((4,), 951, dtype=torch.long)
    xc4 = torch.cat([x[rows], cc[rows]], 1)
    ref = torch.cat([ou.unet_forward(P, cfg, xc4, t4, ctx_c[rows], plan=plan), ou.unet_forward(P, cfg, xc4, t4, ctx_u[rows], plan=plan)])
    m = build(dev, "parity")
    t = torch.full((B,), 951, dtype=torch.long, device=dev)
    out = {}
    for precision in ("parity", "f16", "bf16"):
        m.set_precision(precision)
        ec, eu = m.forward_cfg(x.to(dev), cc.to(dev), t, ctx_c.to(dev), ctx_u.to(dev), uniform_t=True)
        m.check_f16_range()
        out[precision] = dev2(torch.cat([ec[rows], eu[rows]]), ref)
        # every row of the big batch equals the same row evaluated inside a batch of 64 (other launch plan, other tile -> CU placement)
        lo = 192
        ec2, eu2 = m.forward_cfg(x[lo:].to(dev), cc[lo:].to(dev), t[:64], ctx_c[lo:].to(dev), ctx_u[lo:].to(dev), uniform_t=True)
        d64 = dev2(torch.cat([ec[lo:], eu[lo:]]), torch.cat([ec2, eu2]))
        print(f"[NS32 B=256 CFG, rows {rows}, {precision}] vs CPU oracle: rel-L2 {out[precision][0]:.3e}, max/std {out[precision][1]:.3e}; "
              f"rows 192..255 vs the same samples at B=64: rel-L2 {d64[0]:.3e}, max/std {d64[1]:.3e}")
        # parity: the two launch plans agree to fp32 rounding. Single-product modes: a different split-K order moves fp32 sums by ~1e-7, which
        # flips operand roundings downstream — the two plans' rounding noise is only partly the same realisation (measured f16 5.8e-4,
        # bf16 4.6e-3: of the size of the mode's own deviation from the oracle, as it must be), bounded by twice that deviation
        assert d64[0] < {"parity": 1e-5, "f16": 2e-3, "bf16": 1.5e-2}[precision], (precision, d64)
    assert out["parity"][0] < 1e-3 and out["parity"][1] < 1e-3
    # f16 (the bench headline's mode): inside north_star's 1e-3 as rel-L2 (measured 7.5e-4 .. 7.9e-4, round 5); the max-norm reading is
    # bounded at what is measured (4.0e-3 .. 4.5e-3) + margin, so a regression shows - the mode that meets 1e-3 on BOTH readings is `parity`
    assert out["f16"][0] < 1e-3 and out["f16"][1] < 5.5e-3
    assert out["bf16"][0] < 1.5e-2 and out["bf16"][1] < 8e-2


def test_config5_latent64_batch64_cfg_rows_vs_cpu_oracle(dev, oracle_ns32):
    """BASELINE config 5's per-GPU share (CATCH 512^2 images: 64x64x4 latents, 512 / 8 = 64 per GPU; its ns = 8 fp8 style encoder is
    covered in tests/test_gpu_style.py): the same U-Net on 4x the pixels — 64-pixel rows, attention over 256 tokens, other tile
    geometries and split-K choices than at 32^2. Two rows of the CFG pass against the CPU oracle on those samples."""
    from oracle import unet as ou
    cfg, plan, P = oracle_ns32
    B = 64
    x = prng.normal(5, "c5.x", (B, 4, 64, 64)); cc = prng.normal(5, "c5.cc", (B, 3, 64, 64))
    ctx_c = prng.normal(5, "c5.ctx", (B, 512)); ctx_u = prng.normal(5, "c5.ctxu", (B, 512))
    rows = [0, 63]
    t2 = torch.full((2,), 500, dtype=torch.long)
    xc2 = torch.cat([x[rows], cc[rows]], 1)
    ref = torch.cat([ou.unet_forward(P, cfg, xc2, t2, ctx_c[rows], plan=plan), ou.unet_forward(P, cfg, xc2, t2, ctx_u[rows], plan=plan)])
    m = build(dev, "parity")
    t = torch.full((B,), 500, dtype=torch.long, device=dev)
    out = {}
    for precision in ("parity", "f16", "bf16"):
        m.set_precision(precision)
        ec, eu = m.forward_cfg(x.to(dev), cc.to(dev), t, ctx_c.to(dev), ctx_u.to(dev), uniform_t=True)
        out[precision] = dev2(torch.cat([ec[rows], eu[rows]]), ref)
        print(f"[NS32 U-Net on 64x64x4 latents, B=64 CFG, rows {rows}, {precision}] vs CPU oracle: rel-L2 {out[precision][0]:.3e}, max/std {out[precision][1]:.3e}")
    assert out["parity"][0] < 1e-3 and out["parity"][1] < 1e-3
    # f16 (the bench headline's mode): inside north_star's 1e-3 as rel-L2 (measured 7.5e-4 .. 7.9e-4, round 5); the max-norm reading is
    # bounded at what is measured (4.0e-3 .. 4.5e-3) + margin, so a regression shows - the mode that meets 1e-3 on BOTH readings is `parity`
    assert out["f16"][0] < 1e-3 and out["f16"][1] < 5.5e-3
    assert out["bf16"][0] < 1.5e-2 and out["bf16"][1] < 8e-2


def test_config1_batch1_ddim20_cfg_loop_vs_cpu_oracle(dev, oracle_ns32):
    """BASELINE config 1: NS32, batch 1, DDIM 20 steps, cfg 1.5 (rescale 0.7), eta 0, through LatentDiffusion.sample_log with the
    hipGraph replay — every step on the 16-way split-K kernels — against oracle.ddim.ddim_sample with the oracle U-Net (40 CPU
    forwards). parity mode < 1e-3 on the final latent; the single-product modes' accumulated deviation over the 20 steps is reported."""
    from oracle import ddim as od
    from oracle import unet as ou
    from stedm_amd.latent_diffusion import LatentDiffusion
    cfg, plan, P = oracle_ns32
    xT = prng.normal(1, "c1.xT", (1, 4, 32, 32))
    cc = (prng.normal(2, "c1.layout", (1, 3, 32, 32)) > 0).float()
    ctx = prng.normal(3, "c1.ctx", (1, 512)); ctx_u = prng.normal(4, "c1.ctxu", (1, 512))

    def apply_model(x, t, c):
        return ou.unet_forward(P, cfg, torch.cat([x, c["c_concat"][0]], 1), t, c["c_crossattn"][0], plan=plan)

    ref = od.ddim_sample(apply_model, od.Schedule(), xT, {"c_concat": [cc], "c_crossattn": [ctx]}, 20, 0.0,
                         uncond={"c_concat": [cc], "c_crossattn": [ctx_u]}, scale=1.5)
    unet = build(dev, "parity")
    ld = LatentDiffusion(unet, linear_start=0.0015, linear_end=0.0205, image_size=32, channels=4, conditioning_key="hybrid", loss_type="l1",
                         use_graph=True).to(dev)
    cond = {"c_concat": [cc.to(dev)], "c_crossattn": [ctx.to(dev)]}
    unc = {"c_concat": [cc.to(dev)], "c_crossattn": [ctx_u.to(dev)]}
    res = {}
    for precision in ("parity", "f16", "bf16"):
        unet.set_precision(precision)
        s, inter = ld.sample_log(cond, 1, True, 20, eta=0.0, x_T=xT.to(dev), unconditional_conditioning=unc, unconditional_guidance_scale=1.5,
                                 log_every_t=1000)
        res[precision] = dev2(s, ref)
        print(f"[config 1: NS32 B=1 DDIM-20 cfg 1.5, graph replay, {precision}] final latent vs CPU oracle loop: rel-L2 {res[precision][0]:.3e}, "
              f"max/std {res[precision][1]:.3e}")
        assert bool(torch.isfinite(s).all())
    assert res["parity"][0] < 1e-3 and res["parity"][1] < 1e-3
    # f16 (the bench headline's mode) stays inside the 1e-3 tolerance over the whole loop (measured 2.7e-4); bf16 does not (2.3e-3) and is bounded loosely
    assert res["f16"][0] < 1e-3 and res["bf16"][0] < 1e-1


def test_config2_batch64_bf16_train_step_vs_cpu_oracle(dev, oracle_ns32):
    """BASELINE config 2: one NS32 training step at batch 64 in bf16 (the bench's train_step leg). The loss is a batch mean of
    per-sample means, so sample b's dL/dx and dL/dcontext depend on sample b alone (x 1/B): 4 rows are checked against autograd over
    the oracle on those 4 samples; every parameter gradient's norm is checked against a batch-64 parity-mode run of the same step
    (itself pinned to the reference's autograd by F14 at batch 2), and the loss against the oracle's forward on the 4 rows' share."""
    from oracle import train as otrain
    from stedm_amd.train import UNetTrainer
    cfg, plan, P = oracle_ns32
    B = 64
    g = torch.Generator().manual_seed(5)
    x = prng.normal(5, "c2.x", (B, 4, 32, 32)); cc = (prng.normal(5, "c2.cc", (B, 3, 32, 32)) > 0).float()
    ctx = prng.normal(5, "c2.ctx", (B, 512)); tgt = prng.normal(5, "c2.tgt", (B, 4, 32, 32))
    t = torch.randint(0, 1000, (B,), generator=g)
    rows = [0, 21, 42, 63]
    lref, grads, dx_ref, dctx_ref, y_ref = otrain.unet_loss_and_grads(P, cfg, torch.cat([x[rows], cc[rows]], 1), t[rows], ctx[rows], tgt[rows])
    dx_ref, dctx_ref = dx_ref * (4.0 / B), dctx_ref * (4.0 / B)          # oracle mean over 4 samples -> share in a mean over 64
    m = build(dev, "parity")
    stats = {}
    for precision in ("parity", "bf16"):
        m.set_precision(precision)
        tr = UNetTrainer(m)
        loss, dx, dctx = tr.loss_and_backward(x.to(dev), cc.to(dev), t.to(dev), ctx.to(dev), tgt.to(dev))
        per_row = (tgt[rows].to(dev) - tr.forward(x.to(dev), cc.to(dev), t.to(dev), ctx.to(dev))[rows]).abs().mean(dim=[1, 2, 3]).mean()
        stats[precision] = dict(loss=float(loss), rows_loss=float(per_row), dx=dev2(dx[rows], dx_ref), dctx=dev2(dctx[rows], dctx_ref),
                                norms={n: float(p.grad.double().norm()) for n, p in m.named_parameters()})
        print(f"[config 2: NS32 B=64 train step, {precision}] loss {float(loss):.5f}; rows {rows}: loss share {float(per_row):.5f} (oracle {lref:.5f}), "
              f"dL/dx rel-L2 {stats[precision]['dx'][0]:.3e}, dL/dcontext rel-L2 {stats[precision]['dctx'][0]:.3e}")
    p, b = stats["parity"], stats["bf16"]
    assert abs(p["rows_loss"] - lref) < 1e-4 * lref and p["dx"][0] < 1e-3 and p["dctx"][0] < 1e-3
    assert abs(b["rows_loss"] - lref) < 2e-2 * lref and b["dx"][0] < 1e-1 and b["dctx"][0] < 1e-1
    assert abs(b["loss"] - p["loss"]) < 2e-2 * p["loss"]
    nmax = max(p["norms"].values())
    errs = [abs(b["norms"][n] - v) / v for n, v in p["norms"].items() if v > 1e-6 * nmax]
    print(f"[config 2] bf16 vs parity per-tensor gradient norms at batch 64: median {np.median(errs):.2e}, max {max(errs):.2e}")
    assert np.median(errs) < 1e-2 and max(errs) < 0.25


def test_bench_gpus2_self_launch_on_one_device(dev):
    """`python bench.py --gpus 2` starts its two ranks itself (rehearsal: both on cuda:0, gloo in RCCL's place): real model, real
    steps, max-over-ranks timing, all-gather of the samples, ONE line from rank 0 that says n_gpus 2 and counts both ranks' steps — and the
    data-parallel training step on the same ranks (overlapped bucketed gradient all-reduce, weights equal across ranks afterwards)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(STEDM_BENCH_ONE_DEVICE="1", STEDM_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--no-parity-leg", "--no-e2e-leg", "--batch", "16"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["config"]["global_batch"] == 32 and line["value"] > 0
    assert abs(line["value"] - 2 * 3 / (line["ms_per_step"] * 3e-3)) < 1e-2 * line["value"]
    assert len(line["ms_per_step_by_rank"]) == 2 and max(line["ms_per_step_by_rank"]) <= line["ms_per_step"] + 1e-3
    # the roofline's work figure is the EXECUTED convolution work of one rank's step: 43 x 127.213 GFLOP at batch 64 (the shared encoder at B,
    # the decoder at 2B, the sub-pixel upsample at 4 of 9 taps; at this small batch the skip convolutions are launches of their own) - an
    # accounting slip shows here, not as a better fraction
    rl = line["roofline"]
    assert abs(rl["launches_per_step"] * rl["algorithmic_gflop_per_launch"] - 43 * 127.213 * 16 / 64) < 1.0, rl
    # the training half on the same two ranks: per-rank batch, overlapped bucketed all-reduce, identical weights afterwards
    tl = line["train_step"]
    assert tl["n_gpus"] == 2 and tl["global_batch"] == 32 and len(tl["ms_by_rank"]) == 2 and tl["ms"] >= max(tl["ms_by_rank"]) - 1e-2
    assert tl["overlapped_all_reduces"] >= tl["buckets"] and tl["weights_equal_across_ranks"] is True
    assert abs(tl["samples_per_s"] - 32 / (tl["ms"] * 1e-3)) < 1e-2 * tl["samples_per_s"]


_RCCL_ONE_RANK = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from stedm_amd import parallel as par
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)                      # bench.py's call, with the real backend (RCCL)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
dist.barrier()
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)                            # bench.py's max-over-ranks of the timed region
assert float(t.item()) == 1.25
lo, hi = par.shard_range(6, 0, 1)
local = par.per_sample_normal(1, list(range(lo, hi)), (4, 32, 32)).to(dev)
assert par.all_gather_samples(local, 6) is local                    # the helpers short-cut a one-rank group ...
bufs = [torch.empty_like(local)]
dist.all_gather(bufs, local)                                        # ... so issue THEIR collectives directly: the sample all-gather
assert torch.equal(bufs[0], local)
flat = torch.arange(3 * 1024 + 17, dtype=torch.float32, device=dev)
ref = flat.clone()
works = [dist.all_reduce(flat[o:o + 1024], op=dist.ReduceOp.SUM, async_op=True) for o in range(0, flat.numel(), 1024)]
for w in works:                                                     # in-place bucketed all-reduce over views of the gradient arena
    w.wait()
torch.cuda.synchronize()
assert len(works) == 4 and torch.equal(flat, ref)
dist.barrier()
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK")
"""


def test_rccl_backend_single_rank_collectives(dev):
    """The collectives of the N > 1 path on the REAL backend (`nccl` = RCCL), as far as a one-GPU box allows: a one-rank process group
    created exactly as bench.py creates it, the timing all-reduce, `all_gather_samples` and the bucketed gradient all-reduce on device
    tensors. RCCL refuses two ranks on one device, so the two-rank rehearsals above run over gloo; this one proves the library loads,
    initialises on this image (HSA_ENABLE_IPC_MODE_LEGACY=0) and that every call the path makes is one RCCL accepts. In a child
    process: the process group must not leak into the test session."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK, root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
