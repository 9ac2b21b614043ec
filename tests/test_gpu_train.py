"""GPU tier, training step: the hand-scheduled HIP backward against the REFERENCE's own gradients
(tests/golden/f14_grads_*.npz: reference UNetModel forward + L1 + autograd backward) and against the oracle restatement.
Tolerance: 1e-3 relative (per-tensor L2 norm; sampled entries relative to the tensor's RMS) in parity mode."""
import numpy as np
import pytest
import torch

from stedm_amd.utils import prng
from tests.golden.make_golden_grads import pick_index
from tests.golden.summary import check_summary

pytestmark = pytest.mark.gpu

TINY = dict(image_size=16, in_channels=7, model_channels=32, out_channels=4, num_res_blocks=2,
            attention_resolutions=[32, 16, 8], channel_mult=[1, 2, 4], num_heads=4)
NS32 = dict(image_size=32, in_channels=7, model_channels=128, out_channels=4, num_res_blocks=2,
            attention_resolutions=[32, 16, 8], channel_mult=[1, 4, 8], num_heads=8)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def build(cfg, seed, dev, precision="parity"):
    from stedm_amd.unet import UNetModel
    m = UNetModel(precision=precision, **cfg).eval()
    prng.fill_module_(m, seed=seed)
    return m.to(dev)


def _inputs(tag, cfg, B, hw, seed, dev):
    x = prng.normal(seed, f"unet.{tag}.x", (B, cfg["in_channels"], hw, hw)).to(dev)
    ctx = prng.normal(seed, f"unet.{tag}.ctx", (B, cfg["model_channels"] * 4)).to(dev)
    target = prng.normal(seed, f"unet.{tag}.target", (B, cfg["out_channels"], hw, hw)).to(dev)
    return x, ctx, target


def _check_grads(m, fx, tol, what):
    worst = (0.0, "")
    nmax = max(float(fx[f"g.{name}.norm"]) for name, _ in m.named_parameters())
    for name, p in m.named_parameters():
        assert p.grad is not None, f"{name}: no gradient"
        n = float(fx[f"g.{name}.norm"])
        a = p.grad.double().reshape(-1).cpu()
        if n < 1e-6 * nmax:
            # mathematically zero gradient (a per-channel bias in front of a GroupNorm with one channel per group): both sides hold
            # rounding noise only
            assert float(a.norm()) < 1e-5 * nmax, f"{what} {name}: expected a vanishing gradient"
            continue
        en = abs(float(a.norm()) - n) / (n + 1e-30)
        rms = n / np.sqrt(a.numel())
        ep = float(np.abs(a[torch.from_numpy(pick_index(a.numel()))].numpy() - fx[f"g.{name}.pick"]).max()) / (rms + 1e-30)
        worst = max(worst, (en, name + " (norm)"), (ep / 30, name + " (samples)"))
        assert en <= tol, f"{what} {name}: grad norm off by {en:.2e}"
        assert ep <= 30 * tol, f"{what} {name}: sampled grad entries off by {ep:.2e} of the tensor's rms"
    return worst


@pytest.mark.parametrize("tag,cfg,B,hw,seed", [("tiny", TINY, 2, 16, 6), ("ns32", NS32, 2, 32, 0)])
def test_unet_backward_vs_reference_golden(dev, golden, tag, cfg, B, hw, seed):
    from stedm_amd.train import UNetTrainer
    fx = golden(f"f14_grads_{tag}")
    m = build(cfg, seed, dev)
    tr = UNetTrainer(m)
    x, ctx, target = _inputs(tag, cfg, B, hw, seed, dev)
    t = torch.from_numpy(fx["t"]).to(dev)
    loss, dx, dctx = tr.loss_and_backward(x[:, :4].contiguous(), x[:, 4:].contiguous(), t, ctx, target)
    assert abs(float(loss) - float(fx["loss"])) < 1e-4 * float(fx["loss"])
    err_c = float((dctx.double().cpu() - torch.from_numpy(fx["dctx"]).double()).norm() / torch.from_numpy(fx["dctx"]).double().norm())
    check_summary(dx, fx, "dx", 2e-3, tag)
    worst = _check_grads(m, fx, 1e-3, tag)
    print(f"[{tag}] loss {float(loss):.6f}  dctx rel-L2 {err_c:.2e}  worst param grad {worst[0]:.2e} at {worst[1]}")
    assert err_c < 1e-3


def test_unet_backward_is_bitwise_reproducible(dev):
    from stedm_amd.train import UNetTrainer
    m = build(TINY, 6, dev)
    tr = UNetTrainer(m)
    x, ctx, target = _inputs("tiny", TINY, 2, 16, 6, dev)
    t = torch.tensor([951, 21], device=dev)
    tr.loss_and_backward(x[:, :4].contiguous(), x[:, 4:].contiguous(), t, ctx, target)
    g1 = [p.grad.clone() for p in m.parameters()]
    tr.loss_and_backward(x[:, :4].contiguous(), x[:, 4:].contiguous(), t, ctx, target)
    assert all(torch.equal(a, p.grad) for a, p in zip(g1, m.parameters()))


@pytest.mark.parametrize("precision,tol", [("bf16", 6e-2), ("f16", 3e-2)])
def test_unet_backward_single_product_modes(dev, golden, precision, tol):
    """single-product forward (bf16 / f16 operands) + bf16 single-product backward: reported against the reference gradients"""
    from stedm_amd.train import UNetTrainer
    fx = golden("f14_grads_tiny")
    m = build(TINY, 6, dev, precision)
    tr = UNetTrainer(m)
    x, ctx, target = _inputs("tiny", TINY, 2, 16, 6, dev)
    t = torch.from_numpy(fx["t"]).to(dev)
    loss, dx, dctx = tr.loss_and_backward(x[:, :4].contiguous(), x[:, 4:].contiguous(), t, ctx, target)
    errs = []
    for name, p in m.named_parameters():
        n = float(fx[f"g.{name}.norm"])
        errs.append(abs(float(p.grad.double().norm()) - n) / n)
    print(f"[{precision}] loss {float(loss):.5f} (ref {float(fx['loss']):.5f}); grad-norm error median {np.median(errs):.2e} max {max(errs):.2e}")
    assert abs(float(loss) - float(fx["loss"])) < 2e-2 * float(fx["loss"])
    assert np.median(errs) < tol


def test_adamw_ema_step_vs_oracle(dev):
    """fused multi-tensor AdamW + EMA kernel against the oracle's restatement of torch.optim.AdamW / LitEma (two steps)."""
    from oracle import train as otrain
    from stedm_amd.train import UNetTrainer
    m = build(TINY, 6, dev)
    tr = UNetTrainer(m, lr=1e-3, weight_decay=0.01, ema_decay=0.9999)
    m._prepare()
    tr._alloc_grads()
    params = list(m.parameters())
    ref_p = [p.detach().cpu().clone() for p in params]
    ref_m = [torch.zeros_like(p) for p in ref_p]
    ref_v = [torch.zeros_like(p) for p in ref_p]
    ref_e = [p.clone() for p in ref_p]
    for step in (1, 2):
        for i, p in enumerate(params):
            p.grad.copy_(prng.normal(step, f"g{i}", tuple(p.shape)) * 0.01)
        tr._grads_ready = True
        tr.optimizer_step()
        d = otrain.ema_decay(step)
        for i in range(len(params)):
            otrain.adamw_step(ref_p[i], prng.normal(step, f"g{i}", tuple(ref_p[i].shape)) * 0.01, ref_m[i], ref_v[i], step, 1e-3, weight_decay=0.01)
            otrain.ema_update(ref_e[i], ref_p[i], d)
    for i, p in enumerate(params):
        assert torch.allclose(p.detach().cpu(), ref_p[i], rtol=2e-6, atol=2e-7), i
    ema = {id(p): e for p, e in zip(tr._opt["params"], tr.ema_parameters())}
    for i, p in enumerate(params):
        assert torch.allclose(ema[id(p)].cpu(), ref_e[i], rtol=2e-6, atol=2e-7), i


def test_train_steps_reduce_the_loss(dev):
    """a few optimizer steps on a fixed batch: the loss goes down and the packed weights follow the updated parameters"""
    from stedm_amd.train import UNetTrainer
    m = build(TINY, 6, dev, "bf16")
    tr = UNetTrainer(m, lr=2e-4, weight_decay=0.0)
    x, ctx, target = _inputs("tiny", TINY, 2, 16, 6, dev)
    t = torch.tensor([951, 21], device=dev)
    losses = [float(tr.train_step(x[:, :4].contiguous(), x[:, 4:].contiguous(), t, ctx, target)) for _ in range(6)]
    print("losses", ["%.4f" % v for v in losses])
    assert losses[-1] < losses[0] - 0.01


@pytest.mark.parametrize("precision", ["bf16", "f16"])
def test_optimizer_writes_the_weight_packs_itself(dev, precision):
    """stedm_adamw_ema_pack: AdamW + EMA over the convolution weights that also refreshes their fragment-order packs (forward order and the
    flipped / transposed dgrad order, 32x32x16 and 16x16x32 forms). Against the separate launches (AdamW, then stedm_pack_frag_multi before the
    next forward / backward) on the same model and batch: losses of four steps, parameters, EMA shadows and every pack bit for bit; in-place
    edits of a weight between two steps still re-pack it (the freshness mark follows the parameters' versions and UNetModel.invalidate())."""
    from stedm_amd.train import UNetTrainer
    cfg = dict(image_size=16, in_channels=7, model_channels=64, out_channels=4, num_res_blocks=1, attention_resolutions=[32, 16, 8], channel_mult=[1, 4, 8],
               num_heads=4)
    runs = []
    for fuse in (True, False):
        m = build(cfg, 6, dev, precision)
        tr = UNetTrainer(m, lr=2e-4, weight_decay=0.01)
        tr.fuse_packs = fuse
        x, ctx, target = _inputs("fuse", cfg, 2, 16, 6, dev)
        t = torch.tensor([951, 21], device=dev)
        losses = []
        for step in range(4):
            if step == 2:      # an in-place edit (version bump): the packs of this weight must come from the parameter again
                with torch.no_grad():
                    m.input_blocks[1][0].in_layers[2].weight.mul_(0.5)
            if step == 3:      # an edit through .data moves no version: UNetModel.invalidate() is the contract for those
                m.output_blocks[0][0].out_layers[3].weight.data.mul_(2.0)
                m.invalidate()
            losses.append(float(tr.train_step(x[:, :4].contiguous(), x[:, 4:].contiguous(), t, ctx, target)))
        fu = getattr(tr, "_fused", None)
        if fuse:
            assert fu is not None and fu["n"] >= 8, "no convolution weight took the fused path"
            assert all(len(pl._fused) > 0 for pl in fu["plans"]) and len(fu["plans"]) == 2
        else:
            assert fu is None
        m._prepare()           # the forward's packs of the final weights (fused run: only what the optimizer did not write itself)
        if not fuse:
            tr._dplan.run()    # ... and the backward's: the fused run holds them already
        packs = [it[8].clone() for pl in (m._plan, tr._dplan) for it in pl.items]
        runs.append((losses, [p.detach().clone() for p in m.parameters()], [e.clone() for e in tr.ema_parameters()], packs))
    (la, pa, ea, ka), (lb, pb, eb, kb) = runs
    print("losses", ["%.5f" % v for v in la])
    assert la == lb, (la, lb)
    for i, (a, b) in enumerate(zip(pa, pb)):
        assert torch.equal(a, b), f"parameter {i}"
    for i, (a, b) in enumerate(zip(ea, eb)):
        assert torch.equal(a, b), f"ema {i}"
    assert len(ka) == len(kb) and all(torch.equal(a, b) for a, b in zip(ka, kb))


@pytest.mark.parametrize("bucket_mb", [1, 32])
def test_optimizer_inside_the_backward_equals_the_pass_after_it(dev, bucket_mb):
    """Single rank, no accumulation: the optimizer pass of a run of parameters starts on a side stream as soon as the backward has written
    the last of their gradients (UNetTrainer.overlap_optimizer). Same kernels on the same values: losses of five steps, parameters, EMA
    shadows, optimizer moments and every weight pack bit for bit against the pass after the backward; short runs (1 MB) fire many times."""
    from stedm_amd.train import UNetTrainer
    cfg = dict(image_size=16, in_channels=7, model_channels=64, out_channels=4, num_res_blocks=1, attention_resolutions=[32, 16, 8], channel_mult=[1, 4, 8],
               num_heads=4)
    runs = []
    for overlap in (True, False):
        m = build(cfg, 6, dev, "bf16")
        tr = UNetTrainer(m, lr=2e-4, weight_decay=0.01)
        tr.overlap_optimizer = overlap
        tr.opt_bucket_mb = bucket_mb
        x, ctx, target = _inputs("fuse", cfg, 2, 16, 6, dev)
        t = torch.tensor([951, 21], device=dev)
        losses = [float(tr.train_step(x[:, :4].contiguous(), x[:, 4:].contiguous(), t, ctx, target)) for _ in range(5)]
        torch.cuda.synchronize()
        if overlap:
            nruns = len(tr._opt_sched().bounds)
            assert tr.overlap_opt_fires > 0 and nruns >= (8 if bucket_mb == 1 else 1), (tr.overlap_opt_fires, nruns)
            print(f"optimizer runs per step: {nruns}, started inside the backward over 5 steps: {tr.overlap_opt_fires}")
        else:
            assert tr.overlap_opt_fires == 0
        m._prepare()
        packs = [it[8].clone() for pl in (m._plan, tr._dplan) for it in pl.items]
        runs.append((losses, [p.detach().clone() for p in m.parameters()], [e.clone() for e in tr.ema_parameters()],
                     [v.clone() for v in tr._opt["m"]] + [v.clone() for v in tr._opt["v"]], packs))
    (la, pa, ea, ma, ka), (lb, pb, eb, mb, kb) = runs
    assert la == lb, (la, lb)
    for name, xa, xb in (("parameter", pa, pb), ("ema", ea, eb), ("moment", ma, mb), ("pack", ka, kb)):
        assert len(xa) == len(xb)
        for i, (a, b) in enumerate(zip(xa, xb)):
            assert torch.equal(a, b), f"{name} {i}"


@pytest.mark.parametrize("precision,overlap", [("bf16", False), ("f16", False), ("bf16", True), ("parity", False)])
def test_captured_train_step_equals_the_eager_step_bitwise(dev, precision, overlap):
    """UNetTrainer.train_step_graphed: three eager steps, then the whole step (forward, L1, backward, AdamW + EMA + re-pack) captured once and
    replayed as one hipGraph launch with a different batch every step. Against train_step() on a second copy of the model: the loss of each
    of nine steps, every parameter, EMA shadow, optimizer moment and weight pack bit for bit (the optimizer kernels read this step's bias
    corrections / EMA decay / learning rate from the device schedule, built with the host arithmetic of the eager launch — in bf16 with a
    window of 3 rows, so that the schedule is re-filled between replays); a learning-rate
    change between two replays reaches the device schedule; an eager step in between (a stranger moving the parameters) drops the graph and
    the following calls re-capture; with the optimizer overlapped on a side stream the fork / join is part of the graph."""
    from stedm_amd.train import UNetTrainer
    cfg = dict(image_size=16, in_channels=7, model_channels=64, out_channels=4, num_res_blocks=1, attention_resolutions=[32, 16, 8], channel_mult=[1, 4, 8],
               num_heads=4)
    runs = []
    for graphed in (True, False):
        m = build(cfg, 6, dev, precision)
        tr = UNetTrainer(m, lr=2e-4, weight_decay=0.01)
        tr.overlap_optimizer = overlap
        tr.opt_bucket_mb = 1
        if precision == "bf16":
            tr.GRAPH_WINDOW = 3      # (the device schedule is re-filled every other replay: rows 1, 2 of a window, then the next window)
        losses, replays = [], 0
        for step in range(9):
            x, ctx, target = _inputs(f"cap{step}", cfg, 2, 16, 6 + step, dev)
            t = torch.tensor([951 - 7 * step, 21 + step], device=dev)
            if step == 5:
                tr.lr = 1e-4
            a = (x[:, :4].contiguous(), x[:, 4:].contiguous(), t, ctx, target)
            if graphed and step != 6:
                losses.append(float(tr.train_step_graphed(*a)))
                replays += int(getattr(tr, "_graph", None) is not None)
            else:
                losses.append(float(tr.train_step(*a)))      # step 6 of the graphed run: eager, behind the graph's back
        torch.cuda.synchronize()
        if graphed:
            # steps 0-2 eager, 3-5 replayed, 6 eager (drops the graph), 7-8 eager again (warm-up of the re-capture)
            assert replays == 3 and tr._graph is None and tr.step_count == 9 and tr.ema_updates == 9, (replays, tr.step_count, tr.ema_updates)
            for step in range(9, 13):
                x, ctx, target = _inputs(f"cap{step}", cfg, 2, 16, 6 + step, dev)
                losses.append(float(tr.train_step_graphed(x[:, :4].contiguous(), x[:, 4:].contiguous(), torch.tensor([5 + step, 700], device=dev), ctx, target)))
            assert tr._graph is not None and tr.step_count == 13
        else:
            for step in range(9, 13):
                x, ctx, target = _inputs(f"cap{step}", cfg, 2, 16, 6 + step, dev)
                losses.append(float(tr.train_step(x[:, :4].contiguous(), x[:, 4:].contiguous(), torch.tensor([5 + step, 700], device=dev), ctx, target)))
        m._prepare()
        packs = [it[8].clone() for pl in (m._plan, tr._dplan) for it in pl.items] if precision != "parity" else []
        runs.append((losses, [p.detach().clone() for p in m.parameters()], [e.clone() for e in tr.ema_parameters()],
                     [v.clone() for v in tr._opt["m"]] + [v.clone() for v in tr._opt["v"]], packs))
    (la, pa, ea, ma, ka), (lb, pb, eb, mb, kb) = runs
    print("losses", ["%.5f" % v for v in la])
    assert la == lb, (la, lb)
    for name, xa, xb in (("parameter", pa, pb), ("ema", ea, eb), ("moment", ma, mb), ("pack", ka, kb)):
        assert len(xa) == len(xb)
        for i, (a, b) in enumerate(zip(xa, xb)):
            assert torch.equal(a, b), f"{name} {i}"


def test_captured_train_step_device_schedule_rows_are_the_eager_scalars(dev):
    """the rows of the device schedule against the scalars stedm_adamw_ema computes on the host (C float betas, double powers, sqrtf)"""
    import ctypes
    from stedm_amd.train import UNetTrainer
    tr = UNetTrainer(build(TINY, 6, dev, "bf16"), lr=3e-4, betas=(0.9, 0.999), ema_decay=0.9999)
    rows = tr._sched_rows(1, 1, 3000)
    libm = ctypes.CDLL("libm.so.6")
    libm.pow.restype = ctypes.c_double
    libm.pow.argtypes = [ctypes.c_double, ctypes.c_double]
    b1, b2 = ctypes.c_float(0.9).value, ctypes.c_float(0.999).value
    for i in (0, 1, 2, 9, 10, 99, 1000, 2999):
        s = 1 + i
        assert rows[i, 0] == np.float32(1.0 - libm.pow(b1, float(s)))
        assert rows[i, 1] == np.sqrt(np.float32(1.0 - libm.pow(b2, float(s))))
        assert rows[i, 2] == np.float32(min(0.9999, (1 + s) / (10 + s))) and rows[i, 3] == np.float32(3e-4)


def test_spatial_rescaler_weight_gradient_vs_oracle(dev):
    """cond_stage_trainable: channel_mapper.weight gradient from the c_concat slice of the U-Net's input gradient, against autograd
    over the oracle's restatement of SpatialRescaler.forward (encoders/modules.py:123-130)."""
    from oracle import style as ostyle
    from stedm_amd.style import SpatialRescaler
    m = SpatialRescaler(n_stages=2, in_channels=2, out_channels=3).to(dev)
    prng.fill_module_(m, seed=9)
    x = prng.uniform(9, "resc.x", (3, 2, 64, 64))
    d = prng.normal(9, "resc.d", (3, 3, 16, 16))
    w = m.channel_mapper.weight.detach().cpu().clone().requires_grad_(True)
    with torch.enable_grad():
        (ostyle.spatial_rescaler.__wrapped__(x, w, 2) * d).sum().backward()
    g = m.backward(x.to(dev), d.to(dev))
    assert torch.allclose(g.cpu(), w.grad, rtol=1e-5, atol=1e-5)
    g2 = m.backward(x.to(dev), d.to(dev), accumulate=True)
    assert torch.allclose(g2.cpu(), 2 * w.grad, rtol=1e-5, atol=1e-5)


def test_axpby_with_beta_zero_does_not_read_the_destination(dev):
    """stedm_axpby_f32 opens an accumulation window with beta = 0 on a torch.empty arena: whatever the memory held (a NaN from a freed test
    tensor turned every gradient of the window into NaN) must not enter the result."""
    from stedm_amd import ops
    x = torch.randn(4096, device=dev)
    y = torch.full((4096,), float("nan"), device=dev)
    ops.axpby(x, y, 0.5, 0.0)
    assert torch.equal(y, 0.5 * x)
    ops.axpby(x, y, 0.25, 1.0)
    assert torch.allclose(y, 0.75 * x, rtol=1e-6, atol=0)


def test_gradient_accumulation_equals_the_full_batch(dev, golden):
    """accumulate_grad_batches = 2 over the two samples of the F14 case reproduces the full-batch gradient of the reference (the L1 loss is a
    mean of per-sample means); the optimizer runs only on the second micro-batch."""
    from stedm_amd.train import UNetTrainer
    fx = golden("f14_grads_tiny")
    m = build(TINY, 6, dev)
    tr = UNetTrainer(m, lr=0.0, weight_decay=0.0, accumulate_grad_batches=2)
    x, ctx, target = _inputs("tiny", TINY, 2, 16, 6, dev)
    t = torch.from_numpy(fx["t"]).to(dev)
    for i in range(2):
        sl = slice(i, i + 1)
        tr.train_step(x[sl, :4].contiguous(), x[sl, 4:].contiguous(), t[sl], ctx[sl].contiguous(), target[sl].contiguous())
        assert tr.step_count == i          # 0 after the first micro-batch, 1 after the second
    _check_grads(m, fx, 1e-3, "accumulated")


@pytest.mark.parametrize("B,H,W,cin,cout", [(3, 8, 8, 128, 64), (2, 16, 16, 256, 128), (2, 32, 32, 128, 64), (5, 16, 8, 128, 192), (64, 8, 8, 256, 128),
                                             (2, 64, 64, 128, 128), (1, 5, 64, 128, 64), (16, 8, 8, 2048, 1024)])
def test_direct_wgrad3x3_kernel(dev, B, H, W, cin, cout):
    """stedm_wgrad3x3 (both operands from the NHWC bf16 planes, transposed LDS reads, split over pixel units) + the fixed-order
    reduce/scatter against conv2d's weight gradient computed in float64 from the same bf16-rounded operands."""
    from stedm_amd import ops
    prec = ops.Precision.parse("bf16")
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.randn(B, H, W, cin, generator=g).bfloat16()
    dy = (torch.randn(B, H, W, cout, generator=g) * 0.1).bfloat16()
    ks = ops.wgrad3x3_plan(B, H, W, cin, cout)
    assert ks >= 1
    part = torch.empty((ks * 9 * cin * cout,), dtype=torch.float32, device=dev)
    ops.wgrad3x3(x.view(torch.int16).to(dev), dy.view(torch.int16).to(dev), part, prec)
    grad = torch.zeros((cout, cin, 3, 3), dtype=torch.float32, device=dev)
    ops.wgrad_to_oihw(part, grad, cin, cout, False, ks)
    ref = torch.nn.grad.conv2d_weight(x.double().permute(0, 3, 1, 2), (cout, cin, 3, 3), dy.double().permute(0, 3, 1, 2), padding=1)
    err = float((grad.cpu().double() - ref).abs().max() / ref.abs().max())
    print(f"wgrad3x3 B={B} {H}x{W} {cin}->{cout}: ksplit {ks}, max err / max {err:.2e}")
    assert err < 1e-5
    assert ops.wgrad3x3_plan(B, H, 12, cin, cout) == 0 and ops.wgrad3x3_plan(B, H, W, 96, cout) == 0      # unsupported shapes are declined
    # the form the training step runs (round 4): slices in the parameter's own OIHW order, 16-B stores; one slice lands in the gradient itself,
    # several are added by a streaming pass — the same sums in the same order as the transposing reduce above: bit for bit
    part2 = torch.full((ks * 9 * cin * cout,), float("nan"), dtype=torch.float32, device=dev)
    grad2 = torch.full((cout, cin, 3, 3), float("nan"), dtype=torch.float32, device=dev)
    if ks == 1:
        ops.wgrad3x3_oihw(x.view(torch.int16).to(dev), dy.view(torch.int16).to(dev), grad2, prec)
    else:
        ops.wgrad3x3_oihw(x.view(torch.int16).to(dev), dy.view(torch.int16).to(dev), part2, prec)
        ops.sum_planes(part2, grad2, ks)
    assert torch.equal(grad2, grad), f"OIHW-order slices differ from the [tap][ci][co] form (ksplit {ks})"


def test_direct_wgrad_path_equals_im2col_gemm_path(dev, golden):
    """NS32 U-Net, bf16 single-product training step: weight gradients from the direct 3x3 kernel (incl. the Upsample convs over the
    materialised nearest-2x plane) equal those of the im2col + GEMM form — same bf16 operands, fp32 accumulation in another order."""
    from stedm_amd.train import UNetTrainer
    fx = golden("f14_grads_ns32")
    m = build(NS32, 0, dev, "bf16")
    x, ctx, target = _inputs("ns32", NS32, 2, 32, 0, dev)
    t = torch.from_numpy(fx["t"]).to(dev)
    grads = []
    for direct in (True, False):
        tr = UNetTrainer(m)
        tr.direct_wgrad = direct
        tr.loss_and_backward(x[:, :4].contiguous(), x[:, 4:].contiguous(), t, ctx, target)
        grads.append({n: p.grad.clone() for n, p in m.named_parameters()})
        for p in m.parameters():
            p.grad = None
    worst = 0.0
    for n in grads[0]:
        a, b = grads[0][n].double(), grads[1][n].double()
        worst = max(worst, float((a - b).norm() / (b.norm() + 1e-30)) if float(b.norm()) > 1e-12 else 0.0)
    print(f"direct vs GEMM wgrad: worst relative difference over all parameters {worst:.2e}")
    assert worst < 1e-4


def test_latent_diffusion_training_surface(dev):
    """LatentDiffusion.q_sample (bit-exact vs the oracle's fp32 arithmetic), p_losses (forward value) and p_losses_backward /
    training_step_hip (ddpm.py:277-280, 1015-1048, 345-358) on the module surface, incl. the cond stage's channel-mapper gradient."""
    from oracle import ddim as oddim
    from oracle import style as ostyle
    from oracle import train as otrain
    from oracle import unet as ounet
    from stedm_amd.latent_diffusion import LatentDiffusion
    from stedm_amd.style import SpatialRescaler
    unet = build(TINY, 6, dev)
    resc = SpatialRescaler(n_stages=2, in_channels=2, out_channels=3)
    prng.fill_module_(resc, seed=9)
    ld = LatentDiffusion(unet, linear_start=0.0015, linear_end=0.0205, loss_type="l1", image_size=16, channels=4, conditioning_key="hybrid",
                         cond_stage_config=resc).to(dev)
    B = 2
    x0 = prng.normal(21, "ld.x0", (B, 4, 16, 16)); noise = prng.normal(21, "ld.noise", (B, 4, 16, 16))
    layout = prng.uniform(21, "ld.layout", (B, 2, 64, 64)); ctx = prng.normal(21, "ld.ctx", (B, 128))
    t = torch.tensor([951, 21], dtype=torch.long)
    sched = oddim.Schedule()
    xq_ref = oddim.q_sample(sched, x0, t, noise)
    xq = ld.q_sample(x0.to(dev), t.to(dev), noise.to(dev))
    assert torch.equal(xq.cpu(), xq_ref)                         # same fp32 products and sum
    # oracle: rescaler -> concat -> U-Net -> L1, under autograd
    ocfg = ounet.UNetConfig(image_size=16, in_channels=7, model_channels=32, out_channels=4, channel_mult=(1, 2, 4), num_heads=4)
    P = prng.fill_state_dict(ounet.build_plan(ocfg).shapes, 6)
    wm = resc.channel_mapper.weight.detach().cpu().clone().requires_grad_(True)
    with torch.enable_grad():
        cc_ref = ostyle.spatial_rescaler.__wrapped__(layout, wm, 2)
    loss_ref, grads, dx_ref, dctx_ref, _ = otrain.unet_loss_and_grads(P, ocfg, torch.cat([xq_ref, cc_ref.detach()], 1), t, ctx, noise)
    with torch.enable_grad():
        (cc_ref * dx_ref[:, 4:]).sum().backward()
    cc = ld.get_learned_conditioning(layout.to(dev))
    cond = {"c_concat": [cc], "c_crossattn": [ctx.to(dev)]}
    lv, _ = ld.p_losses(x0.to(dev), cond, t.to(dev), noise.to(dev))
    assert abs(float(lv) - loss_ref) < 1e-4 * loss_ref
    loss, _, dx, dctx = ld.p_losses_backward(x0.to(dev), cond, t.to(dev), noise.to(dev), cond_input=layout.to(dev))
    assert abs(float(loss) - loss_ref) < 1e-4 * loss_ref
    assert float((dctx.cpu() - dctx_ref).norm() / dctx_ref.norm()) < 1e-3
    assert float((resc.channel_mapper.weight.grad.cpu() - wm.grad).norm() / wm.grad.norm()) < 1e-3
    gw = unet.input_blocks[1][0].in_layers[2].weight.grad.cpu()
    assert float((gw - grads["input_blocks.1.0.in_layers.2.weight"]).norm() / grads["input_blocks.1.0.in_layers.2.weight"].norm()) < 1e-3
    before = unet.out[2].weight.detach().clone()
    ld.training_step_hip(x0.to(dev), cond, t.to(dev), noise.to(dev))
    assert not torch.equal(before, unet.out[2].weight.detach())   # the optimizer ran
    # cond_stage_trainable (conf/diffusion/ldm_based.yaml:13): the channel mapper joins the same AdamW instance, as in the reference's
    # configure_optimizers; the aggregation block does not
    ld.cond_stage_trainable = True
    tr = ld.configure_trainer(lr=1e-3, weight_decay=0.0)
    assert [id(p) for p in tr.extra_params] == [id(resc.channel_mapper.weight)]
    wm0 = resc.channel_mapper.weight.detach().clone()
    with pytest.raises(ValueError):
        ld.training_step_hip(x0.to(dev), cond, t.to(dev), noise.to(dev))
    ld.training_step_hip(x0.to(dev), cond, t.to(dev), noise.to(dev), cond_input=layout.to(dev))
    step = (resc.channel_mapper.weight.detach() - wm0).abs()
    assert float(step.max()) > 5e-4 and float(step.max()) < 1.1e-3            # first AdamW step: |delta| = lr for every entry with a gradient


@pytest.mark.parametrize("B,H,W", [(3, 32, 16), (1, 16, 64)])
def test_unet_backward_odd_shapes_vs_oracle(dev, B, H, W):
    """ragged cases: odd batch sizes and non-square latents (pixel counts that are not multiples of 64 exercise the padded K of the
    wgrad GEMMs, the stride-2 zero-insert and the 2x2 sums on rectangles) against autograd over the oracle."""
    from oracle import train as otrain
    from oracle import unet as ounet
    from stedm_amd.train import UNetTrainer
    cfg = dict(image_size=16, in_channels=7, model_channels=32, out_channels=4, num_res_blocks=1, attention_resolutions=[32, 16, 8],
               channel_mult=[1, 2, 2], num_heads=4)
    m = build(cfg, 31, dev)
    ocfg = ounet.UNetConfig(image_size=16, in_channels=7, model_channels=32, out_channels=4, num_res_blocks=1, channel_mult=(1, 2, 2), num_heads=4)
    P = prng.fill_state_dict(ounet.build_plan(ocfg).shapes, 31)
    x = prng.normal(31, "odd.x", (B, 7, H, W)); ctx = prng.normal(31, "odd.ctx", (B, 128)); target = prng.normal(31, "odd.t", (B, 4, H, W))
    t = torch.tensor(([951, 21, 500] * B)[:B], dtype=torch.long)
    loss_ref, grads, dx_ref, dctx_ref, _ = otrain.unet_loss_and_grads(P, ocfg, x, t, ctx, target)
    tr = UNetTrainer(m)
    loss, dx, dctx = tr.loss_and_backward(x[:, :4].contiguous().to(dev), x[:, 4:].contiguous().to(dev), t.to(dev), ctx.to(dev), target.to(dev))
    assert abs(float(loss) - loss_ref) < 1e-4 * loss_ref
    assert float((dx.cpu() - dx_ref).norm() / dx_ref.norm()) < 1e-3
    assert float((dctx.cpu() - dctx_ref).norm() / dctx_ref.norm()) < 1e-3
    gmax = max(float(g.norm()) for g in grads.values())
    for n, p in m.named_parameters():
        gn = float(grads[n].norm())
        if gn > 1e-6 * gmax:
            assert float((p.grad.cpu() - grads[n]).norm()) / gn < 1e-3, n


@pytest.mark.parametrize("depth,B", [(1, 2), (2, 3)])
def test_unet_backward_with_spatial_transformer_vs_oracle(dev, depth, B):
    """`use_spatial_transformer=True` (north_star names the SpatialTransformer; attention.py:218-261, routed without context as the
    reference's TimestepEmbedSequential does): training backward through GroupNorm -> proj_in -> [LN -> self-attn -> +x; LN -> self-attn
    -> +x; LN -> GEGLU FF -> +x] x depth -> proj_out (+x) against autograd over the oracle — every parameter gradient of the U-Net
    (the transformer's 20+ tensors included), dL/dx and dL/dcontext."""
    from oracle import train as otrain
    from oracle import unet as ounet
    from stedm_amd.train import UNetTrainer
    cfg = dict(image_size=16, in_channels=7, model_channels=32, out_channels=4, num_res_blocks=1, attention_resolutions=[32, 16, 8],
               channel_mult=[1, 2, 4], num_heads=4, use_spatial_transformer=True, transformer_depth=depth, context_dim=128)
    m = build(cfg, 41, dev)
    ocfg = ounet.UNetConfig(image_size=16, in_channels=7, model_channels=32, out_channels=4, num_res_blocks=1, channel_mult=(1, 2, 4), num_heads=4,
                            use_spatial_transformer=True, transformer_depth=depth, context_dim=128)
    P = prng.fill_state_dict(ounet.build_plan(ocfg).shapes, 41)
    assert set(P) == set(m.state_dict())
    x = prng.normal(41, "st.x", (B, 7, 16, 16)); ctx = prng.normal(41, "st.ctx", (B, 128)); target = prng.normal(41, "st.t", (B, 4, 16, 16))
    t = torch.tensor(([951, 21, 500] * B)[:B], dtype=torch.long)
    loss_ref, grads, dx_ref, dctx_ref, _ = otrain.unet_loss_and_grads(P, ocfg, x, t, ctx, target)
    tr = UNetTrainer(m)
    loss, dx, dctx = tr.loss_and_backward(x[:, :4].contiguous().to(dev), x[:, 4:].contiguous().to(dev), t.to(dev), ctx.to(dev), target.to(dev))
    assert abs(float(loss) - loss_ref) < 1e-4 * loss_ref
    assert float((dx.cpu() - dx_ref).norm() / dx_ref.norm()) < 1e-3
    assert float((dctx.cpu() - dctx_ref).norm() / dctx_ref.norm()) < 1e-3
    gmax = max(float(g.norm()) for g in grads.values())
    worst = (0.0, "")
    n_st = 0
    for n, p in m.named_parameters():
        gn = float(grads[n].norm())
        if "transformer_blocks" in n or ".proj_in." in n or ".proj_out." in n:
            n_st += 1
        if gn > 1e-6 * gmax:
            e = float((p.grad.cpu() - grads[n]).norm()) / gn
            worst = max(worst, (e, n))
    print(f"[U-Net + SpatialTransformer depth {depth}] worst parameter-gradient rel err {worst[0]:.2e} ({worst[1]}); {n_st} transformer tensors")
    assert worst[0] < 1e-3 and n_st >= 20
    # the fast mode (bf16 operands in forward and backward) through the same path: finite, close in norm; and a whole optimizer step runs
    mb = build(cfg, 41, dev, "bf16")
    trb = UNetTrainer(mb, lr=1e-4)
    lossb, _, _ = trb.loss_and_backward(x[:, :4].contiguous().to(dev), x[:, 4:].contiguous().to(dev), t.to(dev), ctx.to(dev), target.to(dev))
    assert abs(float(lossb) - loss_ref) < 3e-2 * loss_ref
    errs = [abs(float(p.grad.norm()) - float(grads[n].norm())) / float(grads[n].norm()) for n, p in mb.named_parameters() if float(grads[n].norm()) > 1e-3 * gmax]
    assert all(bool(torch.isfinite(p.grad).all()) for p in mb.parameters()) and float(np.median(errs)) < 3e-2
    w0 = mb.middle_block[2].transformer_blocks[0].attn1.to_q.weight.detach().clone()
    trb.train_step(x[:, :4].contiguous().to(dev), x[:, 4:].contiguous().to(dev), t.to(dev), ctx.to(dev), target.to(dev))
    assert trb.step_count == 1 and not torch.equal(w0, mb.middle_block[2].transformer_blocks[0].attn1.to_q.weight)


def test_spatial_transformer_backward_over_several_steps(dev):
    """The stacked q | k | v filter of _st_bwd is a per-step tensor: its dgrad pack must never enter the persistent pack plan (a stale entry
    under a recycled id() would feed LAST step's — or another attention's — weights to the dgrad through to_q / to_k / to_v from step 2 on).
    bf16 (single-product backward operands), inner = 64 and 128 at the lower levels (3 * inner % 64 == 0: the fragment-order pack path). After
    two optimizer steps with a large learning rate the third backward is compared (a) bit for bit with a FRESH trainer on a copy of the stepped
    weights and (b) with autograd over the oracle on those weights; the plan's item count stays constant from the first backward on."""
    from oracle import train as otrain
    from oracle import unet as ounet
    from stedm_amd.train import UNetTrainer
    cfg = dict(image_size=16, in_channels=7, model_channels=32, out_channels=4, num_res_blocks=1, attention_resolutions=[32, 16, 8],
               channel_mult=[1, 2, 4], num_heads=4, use_spatial_transformer=True, transformer_depth=1, context_dim=128)
    ocfg = ounet.UNetConfig(image_size=16, in_channels=7, model_channels=32, out_channels=4, num_res_blocks=1, channel_mult=(1, 2, 4), num_heads=4,
                            use_spatial_transformer=True, transformer_depth=1, context_dim=128)
    B = 2
    x = prng.normal(43, "st3.x", (B, 7, 16, 16)); ctx = prng.normal(43, "st3.ctx", (B, 128)); target = prng.normal(43, "st3.t", (B, 4, 16, 16))
    t = torch.tensor([951, 21], dtype=torch.long)
    args = (x[:, :4].contiguous().to(dev), x[:, 4:].contiguous().to(dev), t.to(dev), ctx.to(dev), target.to(dev))
    m = build(cfg, 41, dev, "bf16")
    tr = UNetTrainer(m, lr=2e-3, weight_decay=0.0)
    n_items = []
    for _ in range(2):
        tr.train_step(*args)
        n_items.append((len(tr._dplan.items), len(tr._dpacks)))
    tr.loss_and_backward(*args)
    n_items.append((len(tr._dplan.items), len(tr._dpacks)))
    assert n_items[0] == n_items[1] == n_items[2], f"the persistent dgrad pack plan grows from step to step: {n_items}"
    m2 = build(cfg, 41, dev, "bf16")
    m2.load_state_dict(m.state_dict())
    m2.invalidate()
    tr2 = UNetTrainer(m2, lr=2e-3, weight_decay=0.0)
    tr2.loss_and_backward(*args)
    diff = [n for (n, p), q in zip(m.named_parameters(), m2.parameters()) if not torch.equal(p.grad, q.grad)]
    assert not diff, f"step-3 gradients differ from a fresh trainer's on the same weights: {diff[:6]}"
    P = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    _, grads, _, _, _ = otrain.unet_loss_and_grads(P, ocfg, x, t, ctx, target)
    gmax = max(float(g.norm()) for g in grads.values())
    errs = {n: float((p.grad.cpu() - grads[n]).norm()) / float(grads[n].norm()) for n, p in m.named_parameters() if float(grads[n].norm()) > 1e-3 * gmax}
    worst = max(errs.items(), key=lambda kv: kv[1])
    print(f"[ST, step 3, bf16] median rel err {np.median(list(errs.values())):.2e}, worst {worst[1]:.2e} ({worst[0]})")
    assert float(np.median(list(errs.values()))) < 8e-2 and worst[1] < 0.3      # bf16 single-product operands in forward and backward (measured 3.5e-2 / 6e-2)


def _ddp_worker(rank, world, port, q, overlap=True, bucket_mb=256, steps=1, accumulate=1):
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stedm_amd.train import UNetTrainer
    dev = torch.device("cuda:0")
    m = build(TINY, 6, dev)
    tr = UNetTrainer(m, lr=1e-3, weight_decay=0.0, accumulate_grad_batches=accumulate)
    tr.overlap_all_reduce, tr.bucket_mb = overlap, bucket_mb
    n = world * steps * accumulate
    x, ctx, target = _inputs("tiny", TINY, n, 16, 6, dev)
    t = torch.tensor(([951, 21, 500, 7] * n)[:n], device=dev)
    for i in range(steps * accumulate):
        sl = slice(i * world + rank, i * world + rank + 1)                       # each rank trains on its own sample of every micro-batch
        tr.train_step(x[sl, :4].contiguous(), x[sl, 4:].contiguous(), t[sl], ctx[sl].contiguous(), target[sl].contiguous())
    if rank == 0:
        out = {n: p.detach().cpu().numpy() for n, p in m.named_parameters()}       # numpy: no shared-memory handles across the exit
        out["__overlap_fires__"] = np.array([tr.overlap_fires])
        out["__steps__"] = np.array([tr.step_count])
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_step_equals_gradient_accumulation(dev):
    """Data-parallel training path on the GPU: two ranks (one process each, gloo over the device tensors of the gradient arena, the same
    code path RCCL takes on a node) train on one sample each, all-reduce the arena in buckets and step AdamW with the 1/world average;
    the resulting weights equal a single process accumulating the two samples (accumulate_grad_batches = 2)."""
    import os
    import torch.multiprocessing as mp
    from stedm_amd.train import UNetTrainer
    m = build(TINY, 6, dev)
    tr = UNetTrainer(m, lr=1e-3, weight_decay=0.0, accumulate_grad_batches=2)
    x, ctx, target = _inputs("tiny", TINY, 2, 16, 6, dev)
    t = torch.tensor([951, 21], device=dev)
    for i in range(2):
        sl = slice(i, i + 1)
        tr.train_step(x[sl, :4].contiguous(), x[sl, 4:].contiguous(), t[sl], ctx[sl].contiguous(), target[sl].contiguous())
    ref = {n: p.detach().cpu() for n, p in m.named_parameters()}
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctxm.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert int(got["__overlap_fires__"][0]) > 0, "the overlapped all-reduce (the default) did not run on the first step"
    worst = max(float(np.abs(got[n] - ref[n].numpy()).max()) for n in ref)
    print(f"two-rank data-parallel step vs gradient accumulation: max |dw| = {worst:.2e} (lr 1e-3)")
    assert worst < 2e-5


def test_overlapped_gradient_all_reduce_equals_the_plain_one_bitwise(dev):
    """two ranks on the device arena: buckets (1 MB here, so that the tiny U-Net has several) all-reduced the moment the backward has
    produced their last gradient vs the same buckets reduced after the backward: identical weights, bit for bit; and the first
    bucket to fire is the arena's LAST (the backward walks from the output layers to the input)."""
    import os
    import torch.multiprocessing as mp
    ctxm = mp.get_context("spawn")
    # (steps, accumulate): two optimizer steps (the second runs on stepped weights and re-packed operands); and accumulate_grad_batches = 2,
    # where a bucket is first folded into the running mean and the collective runs on the accumulation arena
    for ci, (steps, accumulate) in enumerate([(2, 1), (2, 2)]):
        res = []
        for overlap in (True, False):
            q = ctxm.Queue()
            port = 36500 + (os.getpid() % 2000) + 2 * ci + (1 if overlap else 0)
            procs = [ctxm.Process(target=_ddp_worker, args=(r, 2, port, q, overlap, 1, steps, accumulate)) for r in range(2)]
            for p in procs:
                p.start()
            res.append(q.get(timeout=300))
            for p in procs:
                p.join(timeout=120)
                assert p.exitcode == 0
        fires_on, fires_off = int(res[0].pop("__overlap_fires__")[0]), int(res[1].pop("__overlap_fires__")[0])
        assert int(res[0].pop("__steps__")[0]) == int(res[1].pop("__steps__")[0]) == steps
        # every optimizer step fired at least two buckets from inside its backward (the first step included); the plain path none
        assert fires_on >= 2 * steps and fires_off == 0, (fires_on, fires_off)
        for n in res[0]:
            assert np.array_equal(res[0][n], res[1][n]), (steps, accumulate, n)
    from stedm_amd.train import UNetTrainer
    m = build(TINY, 6, dev)
    tr = UNetTrainer(m)
    tr.bucket_mb = 1
    x, ctx, target = _inputs("tiny", TINY, 2, 16, 6, dev)
    fired = []
    tr.loss_and_backward(x[:, :4].contiguous(), x[:, 4:].contiguous(), torch.tensor([951, 21], device=dev), ctx, target, on_bucket=fired.append)
    nb = len(tr._sched.bounds)
    assert nb >= 3 and sorted(fired) == list(range(nb)) and fired[0] == nb - 1       # each bucket once; the arena's end (output layers) first
    assert fired.index(0) > fired.index(nb - 2)                                        # the embedding Linears at the arena's start complete late


@pytest.mark.parametrize("B,T,heads,ch", [(2, 64, 4, 32), (2, 64, 8, 128), (3, 64, 2, 64), (2, 64, 2, 16), (2, 256, 4, 32), (1, 1024, 2, 64), (3, 100, 2, 16)])
def test_attention_backward_vs_autograd(dev, B, T, heads, ch):
    """QKVAttentionLegacy backward: the fp32-MFMA form (T = 64, head channels a multiple of 32), the LDS-resident VALU form (other T <= 128) and the two-kernel general form (T = 256 of the 64x64 latents, T = 1024 of
    the reference-native 128x128 ones) against autograd over the oracle's restatement."""
    from oracle import unet as ounet
    from stedm_amd import ops
    g = torch.Generator().manual_seed(T + heads)
    qkv = torch.randn(B, T, heads * 3 * ch, generator=g)
    d = torch.randn(B, T, heads * ch, generator=g)
    x = qkv.permute(0, 2, 1).clone().requires_grad_(True)          # oracle layout [B, heads*3*ch, T]
    with torch.enable_grad():
        y = ounet.qkv_attention_legacy(x, heads)                   # [B, heads*ch, T]
        (y * d.permute(0, 2, 1)).sum().backward()
    dq = torch.empty_like(qkv, device=dev)
    ops.attn_legacy_bwd(qkv.to(dev), d.to(dev), dq, heads)
    ref = x.grad.permute(0, 2, 1)
    err = float((dq.cpu() - ref).abs().max() / ref.abs().max())
    assert err < 2e-5, err


def test_unet_backward_reference_native_128_latents_vs_oracle(dev):
    """REF128 (the reference's own configuration: 128x128x3 latents, 6 input / 3 output channels, attention over 1024 tokens): one
    loss + backward at batch 1 against autograd over the oracle — the general-T attention backward, batch-chunked GEMM weight gradients
    (W = 128 is outside the direct kernel) and the padded first / last convolutions at full size."""
    from oracle import train as otrain
    from oracle import unet as ounet
    from stedm_amd.train import UNetTrainer
    cfg = dict(image_size=128, in_channels=6, model_channels=128, out_channels=3, num_res_blocks=2, attention_resolutions=[32, 16, 8],
               channel_mult=[1, 4, 8], num_heads=8)
    m = build(cfg, 12, dev, "bf16")
    ocfg = ounet.UNetConfig(image_size=128, in_channels=6, out_channels=3)
    P = prng.fill_state_dict(ounet.build_plan(ocfg).shapes, 12)
    x = prng.normal(12, "ref.x", (1, 6, 128, 128)); ctx = prng.normal(12, "ref.ctx", (1, 512)); target = prng.normal(12, "ref.t", (1, 3, 128, 128))
    t = torch.tensor([500], dtype=torch.long)
    loss_ref, grads, dx_ref, dctx_ref, _ = otrain.unet_loss_and_grads(P, ocfg, x, t, ctx, target)
    tr = UNetTrainer(m)
    loss, dx, dctx = tr.loss_and_backward(x[:, :3].contiguous().to(dev), x[:, 3:].contiguous().to(dev), t.to(dev), ctx.to(dev), target.to(dev))
    errs = [abs(float(p.grad.double().norm().cpu()) - float(grads[n].double().norm())) / float(grads[n].double().norm())
            for n, p in m.named_parameters() if float(grads[n].norm()) > 1e-6 * max(float(g.norm()) for g in grads.values())]
    print(f"REF128 bf16 training step: loss {float(loss):.5f} (oracle {loss_ref:.5f}); grad-norm error median {np.median(errs):.2e} max {max(errs):.2e}; "
          f"dctx rel {float((dctx.cpu() - dctx_ref).norm() / dctx_ref.norm()):.2e}")
    assert abs(float(loss) - loss_ref) < 2e-2 * loss_ref
    assert np.median(errs) < 2e-2


# ---------------------------------------------------------------------------------------------------- the reference's training seam
def _tiny_ld(dev, seed=6, use_ema=True, trainable=True):
    from stedm_amd.latent_diffusion import LatentDiffusion
    from stedm_amd.style import SpatialRescaler
    unet = build(TINY, seed, dev)
    resc = SpatialRescaler(n_stages=2, in_channels=2, out_channels=3)
    prng.fill_module_(resc, seed=9)
    ld = LatentDiffusion(unet, linear_start=0.0015, linear_end=0.0205, loss_type="l1", image_size=16, channels=4, conditioning_key="hybrid",
                         cond_stage_config=resc, cond_stage_trainable=trainable, use_ema=use_ema).to(dev)
    return ld, unet, resc


def _seam_inputs(dev, B=2, seed=21):
    x0 = prng.normal(seed, "ld.x0", (B, 4, 16, 16)); noise = prng.normal(seed, "ld.noise", (B, 4, 16, 16))
    layout = prng.uniform(seed, "ld.layout", (B, 2, 64, 64)); ctx = prng.normal(seed, "ld.ctx", (B, 128))
    t = torch.tensor(([951, 21, 500, 3] * B)[:B], dtype=torch.long)
    return x0, noise, layout, ctx, t


def test_training_step_hip_with_graph_replay_equals_the_eager_surface(dev):
    """LatentDiffusion.training_step_hip(graph=True) with a frozen cond stage (q_sample and the conditioning stay eager, the U-Net's step is the
    captured graph): losses of seven steps, the U-Net's parameters and the EMA shadows bit for bit against graph=False; with a trainable cond
    stage (its gradient is a host callback between backward and optimizer) the same flag runs the eager step."""
    res = []
    for graph in (True, False):
        ld, unet, resc = _tiny_ld(dev, trainable=False)
        ld.train()
        losses = []
        for step in range(7):
            x0, noise, layout, ctx, t = _seam_inputs(dev, seed=21 + step)
            cond = {"c_concat": [ld.get_learned_conditioning(layout.to(dev))], "c_crossattn": [ctx.to(dev)]}
            losses.append(float(ld.training_step_hip(x0.to(dev), cond, t.to(dev), noise.to(dev), graph=graph)))
        tr = ld._trainer_or_default()
        assert (getattr(tr, "_graph", None) is not None) == graph
        res.append((losses, [p.detach().clone() for p in unet.parameters()], [e.clone() for e in tr.ema_parameters()]))
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    for a, b in zip(res[0][1] + res[0][2], res[1][1] + res[1][2]):
        assert torch.equal(a, b)
    ld, unet, resc = _tiny_ld(dev, trainable=True)
    ld.train()
    x0, noise, layout, ctx, t = _seam_inputs(dev)
    cond = {"c_concat": [ld.get_learned_conditioning(layout.to(dev))], "c_crossattn": [ctx.to(dev)]}
    for _ in range(5):
        ld.training_step_hip(x0.to(dev), cond, t.to(dev), noise.to(dev), cond_input=layout.to(dev), graph=True)
    assert getattr(ld._trainer_or_default(), "_graph", None) is None


def test_p_losses_autograd_bridge_with_torch_adamw(dev):
    """The seam Lightning's automatic optimisation uses (ddpm.py:345-358 -> 1015-1048): in training mode p_losses returns a loss with
    a grad_fn; `loss.backward()` fills `.grad` (checked against autograd over the oracle, incl. the cond stage's channel mapper and the
    gradient handed upstream to the style vector), a second backward ACCUMULATES, zero_grad(set_to_none=True) resets, and
    torch.optim.AdamW steps the parameters; on_train_batch_end then runs LitEma's update (oracle restatement pinned by F13)."""
    from oracle import ddim as oddim
    from oracle import style as ostyle
    from oracle import train as otrain
    from oracle import unet as ounet
    ld, unet, resc = _tiny_ld(dev)
    x0, noise, layout, ctx, t = _seam_inputs(dev)
    ocfg = ounet.UNetConfig(image_size=16, in_channels=7, model_channels=32, out_channels=4, channel_mult=(1, 2, 4), num_heads=4)
    P = prng.fill_state_dict(ounet.build_plan(ocfg).shapes, 6)
    wm = resc.channel_mapper.weight.detach().cpu().clone().requires_grad_(True)
    with torch.enable_grad():
        cc_ref = ostyle.spatial_rescaler.__wrapped__(layout, wm, 2)
    xq_ref = oddim.q_sample(oddim.Schedule(), x0, t, noise)
    loss_ref, grads, dx_ref, dctx_ref, _ = otrain.unet_loss_and_grads(P, ocfg, torch.cat([xq_ref, cc_ref.detach()], 1), t, ctx, noise)
    with torch.enable_grad():
        (cc_ref * dx_ref[:, 4:]).sum().backward()

    ld.train()
    params = list(ld.model.parameters()) + list(ld.cond_stage_model.parameters())       # configure_optimizers, ldm_diffusion.py:224-234
    opt = ld.attach_optimizer(torch.optim.AdamW(params, lr=1e-3))
    ctx_d = ctx.to(dev).requires_grad_(True)                        # stands for the output of an autograd-run style encoder
    with torch.enable_grad():
        ld.cond_stage_trainable = True
        cc = ld.get_learned_conditioning(layout.to(dev))
        loss, ldict = ld.p_losses(x0.to(dev), {"c_concat": [cc], "c_crossattn": [ctx_d]}, t.to(dev), noise.to(dev), cond_input=layout.to(dev))
        assert loss.requires_grad and "train/loss" in ldict
        (loss * 0.5).backward()
    assert abs(float(loss) - loss_ref) < 1e-4 * loss_ref
    assert float((ctx_d.grad.cpu() - 0.5 * dctx_ref).norm() / (0.5 * dctx_ref).norm()) < 1e-3
    with torch.enable_grad():
        loss2, _ = ld.p_losses(x0.to(dev), {"c_concat": [cc], "c_crossattn": [ctx_d]}, t.to(dev), noise.to(dev), cond_input=layout.to(dev))
        (loss2 * 0.5).backward()                                     # second micro-batch of an accumulation window: sums into .grad
    gmax = max(float(g.norm()) for g in grads.values())
    for n, p in unet.named_parameters():
        gn = float(grads[n].norm())
        if gn > 1e-6 * gmax:
            assert float((p.grad.cpu() - grads[n]).norm()) / gn < 1e-3, n
    assert float((resc.channel_mapper.weight.grad.cpu() - wm.grad).norm() / wm.grad.norm()) < 1e-3
    before = {n: p.detach().clone() for n, p in unet.named_parameters()}
    opt.step()
    ld.on_train_batch_end()
    opt.zero_grad(set_to_none=True)
    w = unet.out[2].weight
    assert not torch.equal(before["out.2.weight"], w.detach())
    # torch's AdamW did the update: first step moves every entry with a gradient by ~lr
    assert float((w.detach() - before["out.2.weight"]).abs().max()) < 1.1e-3
    # EMA after one update: shadow = p0 - (1 - d)(p0 - p1), d = (1 + 1) / (10 + 1)
    ema = ld._trainer.ema_named()
    d = otrain.ema_decay(1)
    ref = before["out.2.weight"].cpu().clone()
    otrain.ema_update(ref, w.detach().cpu(), d)
    assert torch.allclose(ema["out.2.weight"].cpu(), ref, rtol=2e-6, atol=1e-7) and ld._trainer.ema_updates == 1
    # the next window starts from None gradients
    with torch.enable_grad():
        loss3, _ = ld.p_losses(x0.to(dev), {"c_concat": [cc], "c_crossattn": [ctx_d]}, t.to(dev), noise.to(dev), cond_input=layout.to(dev))
        loss3.backward()
    assert float(loss3) != float(loss) and all(p.grad is not None for p in params)
    # eval mode / no_grad: forward value only, 'val' prefix (ddpm.py:1021)
    ld.eval()
    lv, dv = ld.p_losses(x0.to(dev), {"c_concat": [cc], "c_crossattn": [ctx_d.detach()]}, t.to(dev), noise.to(dev))
    assert not lv.requires_grad and "val/loss" in dv


class _PoolStage(torch.nn.Module):
    """stand-in first stage for the seam tests (the seam is what is tested, not the autoencoder): 4x4 box mean to 4 channels"""

    def encode(self, x):
        z = torch.nn.functional.avg_pool2d(x, 4)
        return torch.cat([z, z[:, :1]], 1)

    def decode(self, z):
        return torch.nn.functional.interpolate(z[:, :3], scale_factor=4)


# S_ZSS_DM builds the style encoder with num_classes = 512 (s_zss_dm.py:33-38): the U-Net's embedding width must be 4 * 128
MODULE_UNET = dict(image_size=16, in_channels=7, model_channels=128, out_channels=4, num_res_blocks=1, attention_resolutions=[32, 16, 8],
                   channel_mult=[1, 2], num_heads=4)


# conf/style_agg/svit.yaml as shipped (dropout 0.1 / emb_dropout 0.1 are LIVE in training_step: the reference runs the agg block in train mode)
SVIT_YAML = dict(name="svit", patch_size=8, dim=256, depth=6, heads=12, mlp_dim=256, pool="mean", channels=3, dropout=0.1, emb_dropout=0.1, t_dim=256)


def _module_cfg(style_agg=None):
    return {"lr": 1e-3, "cfg_scale": 1.5, "ddim_steps": 4, "eta": 0.0, "data": {"patch_size": 64},
            "style_sampling": {"name": "mp", "num_patches": 2},
            "style_agg": dict(SVIT_YAML) if style_agg is None else style_agg,
            "diffusion": dict(linear_start=0.0015, linear_end=0.0205, timesteps=1000, loss_type="l1", first_stage_key="image",
                              cond_stage_key="segmentation", image_size=16, channels=4, conditioning_key="hybrid", cond_stage_trainable=True,
                              unet_config={"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": dict(MODULE_UNET)},
                              cond_stage_config={"target": "ldm.modules.encoders.modules.SpatialRescaler",
                                                 "params": {"n_stages": 2, "in_channels": 2, "out_channels": 3}})}


def _module_batches(dev, n, B=2):
    out = []
    for i in range(n):
        g = torch.Generator().manual_seed(100 + i)
        img = torch.rand(B, 3, 64, 64, generator=g) * 2 - 1
        seg = torch.nn.functional.one_hot(torch.randint(0, 3, (B, 64, 64), generator=g), 3).permute(0, 3, 1, 2).float()
        sty = torch.rand(B, 2, 3, 64, 64, generator=g) * 2 - 1
        out.append((img.to(dev), seg.to(dev), None, sty.to(dev), torch.arange(B) + i * B))
    return out


def _build_module(dev, style_agg=None):
    from stedm_amd.ldm_module import LDM_Diffusion
    mod = LDM_Diffusion(_module_cfg(style_agg), accumulate_grad_batches=2)
    mod._model.first_stage_model = _PoolStage()
    prng.fill_module_(mod._model.model.diffusion_model, seed=6)
    prng.fill_module_(mod._model.cond_stage_model, seed=9)
    if style_agg is None:
        prng.fill_module_(mod._model.agg_block, seed=51)
    else:       # Swin-V2-T keeps torchvision's initialisation (trunc-normal weights); the aggregation MLP takes the PRNG recipe
        lb = getattr(mod._model.agg_block, "linear_block", None)
        if lb is not None:
            prng.fill_module_(lb, seed=51)
    return mod.to(dev)


def test_ldm_module_training_loop_as_the_reference_drives_it(dev):
    """stedm_amd.ldm_module.LDM_Diffusion driven exactly as Lightning drives modules/ldm_diffusion.py:63-73, 110-115: per batch
    on_train_batch_start -> training_step(batch, batch_idx) -> on_train_batch_end, accumulate_grad_batches = 2 over 4 micro-batches:
    2 AdamW steps, 4 LitEma updates (ddpm.py:369-371 runs after every micro-batch). The weights equal those of the same micro-batches
    pushed through UNetTrainer.train_step by hand (same RNG draws), and the checkpoint written afterwards continues a run (EMA
    history, AdamW moments and bias-correction step) in a freshly built module."""
    from stedm_amd.latent_diffusion import StedmHipError
    batches = _module_batches(dev, 5)
    mod = _build_module(dev)
    mod.train()
    assert mod.configure_optimizers() is None and mod.automatic_optimization is False
    sd_keys = set(mod.state_dict())
    assert "_model.model.diffusion_model.out.2.weight" in sd_keys and "model.model.diffusion_model.out.2.weight" in sd_keys   # both aliases
    torch.manual_seed(1234)
    losses = []
    for idx, b in enumerate(batches[:4]):
        mod.on_train_batch_start(b, idx)
        losses.append(float(mod.training_step(b, idx)))
        mod.on_train_batch_end()
    tr = mod._model._trainer
    assert tr.step_count == 2 and tr.ema_updates == 4 and [id(p) for p in tr.extra_params] == [id(mod._model.cond_stage_model.channel_mapper.weight)]
    assert abs(mod.train_loss() - sum(losses) / 4) < 1e-6

    # the same four micro-batches by hand
    ref = _build_module(dev)
    ref.train()
    m2 = ref._model
    tr2 = m2.configure_trainer(lr=1e-3, accumulate_grad_batches=2)
    torch.manual_seed(1234)
    for b in batches[:4]:
        lb = ref.prepare_batch(b)
        x, c = m2.get_input(lb, "image")[:2]
        t = torch.randint(0, 1000, (x.shape[0],), device=dev).long()
        noise = torch.randn_like(x)
        layout = lb["segmentation"].permute(0, 3, 1, 2).float().contiguous()
        xn = m2.q_sample(x, t, noise)
        tr2.train_step(xn, c["c_concat"][0], t, c["c_crossattn"][0], noise,
                       after_backward=lambda dx, dctx: m2.cond_stage_model.backward(layout, dx[:, 4:].contiguous()))
    for (n, p), (_, q) in zip(mod._model.model.named_parameters(), m2.model.named_parameters()):
        assert torch.equal(p, q), n
    assert torch.equal(mod._model.cond_stage_model.channel_mapper.weight, m2.cond_stage_model.channel_mapper.weight)

    # checkpoint -> fresh module -> one more window; against the uninterrupted run
    ck = mod._model.reference_checkpoint()
    ck = {"state_dict": {k: v.detach().cpu().clone() for k, v in ck["state_dict"].items()},
          "optimizer_states": [{"state": {i: {k: v.detach().cpu().clone() for k, v in s.items()} for i, s in ck["optimizer_states"][0]["state"].items()},
                                "param_groups": ck["optimizer_states"][0]["param_groups"]}]}
    assert int(ck["state_dict"]["_model.model_ema.num_updates"]) == 4
    res = _build_module(dev)
    res.train()
    res._model.load_reference_state_dict(ck)                        # build, load, THEN configure (the usual order)
    res.configure_optimizers()
    for who in (mod, res):
        torch.manual_seed(99)
        for idx, b in enumerate(batches[3:5]):
            who.training_step(b, idx)
            who.on_train_batch_end()
    ta, tb = mod._model._trainer, res._model._trainer
    assert ta.step_count == tb.step_count == 3 and ta.ema_updates == tb.ema_updates == 6
    ea, eb = ta.ema_named(), tb.ema_named()
    for (n, p), (_, q) in zip(mod._model.model.diffusion_model.named_parameters(), res._model.model.diffusion_model.named_parameters()):
        assert torch.equal(p, q), n
        assert torch.equal(ea[n], eb[n]), n
    out = res._model.reference_checkpoint()
    assert int(out["state_dict"]["_model.model_ema.num_updates"]) == 6 and float(out["optimizer_states"][0]["state"][0]["step"]) == 3.0
    # training mode without a first stage must not hand zero latents to the optimizer
    res._model.first_stage_model = None
    with pytest.raises(StedmHipError):
        res.training_step(batches[0], 0)


@pytest.mark.parametrize("agg", ["linear", "mean", "max"])
def test_ldm_module_trains_with_the_default_style_agg(dev, agg):
    """conf/config_diff.yaml:16 `style_agg: linear` (and mean / max): S_ZSS_DM builds Agg_* over Swin-V2-T (s_zss_dm.py:19-27) and get_input
    runs it inside training_step with the module in train mode — torchvision's stochastic depth is live there (gates drawn per forward), the
    embedder itself is not optimised (ldm_diffusion.py:224-234). The loop of the reference (on_train_batch_start -> training_step ->
    on_train_batch_end) must run, step the U-Net, and leave the agg block's weights untouched."""
    batches = _module_batches(dev, 2)
    mod = _build_module(dev, style_agg={"name": agg})
    mod.train()
    agg_before = {n: p.detach().clone() for n, p in mod._model.agg_block.named_parameters()}
    w_before = mod._model.model.diffusion_model.out[2].weight.detach().clone()
    torch.manual_seed(7)
    for idx, b in enumerate(batches):
        mod.on_train_batch_start(b, idx)
        loss = mod.training_step(b, idx)
        mod.on_train_batch_end()
        assert bool(torch.isfinite(loss))
    tr = mod._model._trainer
    assert tr.step_count == 1 and tr.ema_updates == 2
    assert not torch.equal(w_before, mod._model.model.diffusion_model.out[2].weight)
    for n, p in mod._model.agg_block.named_parameters():
        assert torch.equal(p, agg_before[n]), n
    # the style vector of the training step is the train-mode one: stochastic depth drew gates (12 blocks x 2 branches x B * n images)
    emb = mod._model.agg_block.embedder
    assert emb.training and emb._gates is not None and tuple(emb._gates.shape) == (12, 2, 4)
    lb = mod.prepare_batch(batches[0])
    torch.manual_seed(3)
    a = mod._model.get_input(lb, "image")[1]["c_crossattn"][0].clone()
    b2 = mod._model.get_input(lb, "image")[1]["c_crossattn"][0].clone()
    mod.eval()
    e1 = mod._model.get_input(lb, "image")[1]["c_crossattn"][0].clone()
    e2 = mod._model.get_input(lb, "image")[1]["c_crossattn"][0].clone()
    assert torch.equal(e1, e2) and tuple(a.shape) == (2, 512)
    assert not torch.equal(a, b2) or not torch.equal(a, e1)       # (a draw may keep every branch of these 4 images; two draws both doing so is ~1e-3)



def _surface_worker(rank, world, port, q):
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    ld, unet, resc = _tiny_ld(dev)
    ld.configure_trainer(lr=1e-3, weight_decay=0.0)
    x0, noise, layout, ctx, t = _seam_inputs(dev)
    sl = slice(rank, rank + 1)
    cc = ld.get_learned_conditioning(layout[sl].to(dev))
    ld.training_step_hip(x0[sl].to(dev), {"c_concat": [cc], "c_crossattn": [ctx[sl].to(dev)]}, t[sl].to(dev), noise[sl].to(dev), cond_input=layout[sl].to(dev))
    if rank == 0:
        sd = {n: p.detach().cpu().numpy() for n, p in unet.named_parameters()}
        sd["mapper"] = resc.channel_mapper.weight.detach().cpu().numpy()
        q.put(sd)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_through_the_latent_diffusion_surface(dev):
    """training_step_hip under torch.distributed (two processes, gloo over the device arena — the RCCL code path): each rank one sample;
    the bucketed all-reduce + 1/world average inside the step give the weights of one process accumulating both samples."""
    import os
    import torch.multiprocessing as mp
    ld, unet, resc = _tiny_ld(dev)
    ld.configure_trainer(lr=1e-3, weight_decay=0.0, accumulate_grad_batches=2)
    x0, noise, layout, ctx, t = _seam_inputs(dev)
    for i in range(2):
        sl = slice(i, i + 1)
        cc = ld.get_learned_conditioning(layout[sl].to(dev))
        ld.training_step_hip(x0[sl].to(dev), {"c_concat": [cc], "c_crossattn": [ctx[sl].to(dev)]}, t[sl].to(dev), noise[sl].to(dev),
                             cond_input=layout[sl].to(dev))
    ref = {n: p.detach().cpu().numpy() for n, p in unet.named_parameters()}
    ref["mapper"] = resc.channel_mapper.weight.detach().cpu().numpy()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctxm.Process(target=_surface_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    worst = max(float(np.abs(got[n] - ref[n]).max()) for n in ref)
    print(f"two-rank training_step_hip vs accumulation: max |dw| = {worst:.2e} (lr 1e-3)")
    assert worst < 2e-5


@pytest.mark.parametrize("B,H,W,c1,c2,act,use_add,acc", [
    (4, 8, 8, 1024, 0, 1, True, False), (3, 8, 8, 1024, 1024, 1, False, True), (2, 16, 16, 512, 0, 1, True, False),
    (2, 16, 16, 1024, 512, 1, True, False), (2, 32, 32, 128, 0, 1, False, False), (2, 32, 32, 256, 128, 1, True, True),
    (3, 10, 10, 64, 0, 0, True, False), (2, 16, 16, 512, 512, 0, False, False), (1, 64, 64, 128, 0, 1, True, False)])
def test_group_norm_backward_kernels_vs_autograd(dev, B, H, W, c1, c2, act, use_add, acc):
    """stedm_gn_bwd (one-pass LDS-resident form where a channel run of whole groups fits, the statistics / fold / apply chain otherwise:
    1536 and 384 channels, 64 x 64 pixels) against torch autograd of act(GroupNorm32([x1|x2])): dx (+ add, + accumulate), the 16-bit
    planes of dx, dgamma, dbeta. util.py:199-216."""
    import torch.nn.functional as F
    from stedm_amd import ops
    C, G, HW = c1 + c2, 32, H * W
    prec = ops.Precision.parse("parity")
    g = torch.Generator(device="cpu").manual_seed(5)
    x = (torch.randn(B, C, H, W, generator=g, dtype=torch.float64) * 1.5 + 0.3).detach().requires_grad_(True)
    gamma = (1.0 + 0.3 * torch.randn(C, generator=g, dtype=torch.float64)).detach().requires_grad_(True)
    beta = (0.2 * torch.randn(C, generator=g, dtype=torch.float64)).detach().requires_grad_(True)
    dA = torch.randn(B, C, H, W, generator=g, dtype=torch.float64)
    add = torch.randn(B, C, H, W, generator=g, dtype=torch.float64) if use_add else None
    old = torch.randn(B, C, H, W, generator=g, dtype=torch.float64) if acc else None
    with torch.enable_grad():      # (other tests of the session switch autograd off globally)
        y = F.group_norm(x, G, gamma, beta, eps=1e-5)
        if act:
            y = F.silu(y)
        (y * dA).sum().backward()
    want = x.grad + (add if use_add else 0) + (old if acc else 0)

    def nhwc(t):
        return t.permute(0, 2, 3, 1).contiguous().float().to(dev)
    xf = nhwc(x.detach())
    x1 = xf[..., :c1].contiguous()
    x2 = xf[..., c1:].contiguous() if c2 else None
    cs1 = torch.empty(B, ops.gn_chan_nslab(HW), c1, 2, device=dev); ops.gn_chan_stats(x1, cs1)
    cs2 = None
    if c2:
        cs2 = torch.empty(B, ops.gn_chan_nslab(HW), c2, 2, device=dev); ops.gn_chan_stats(x2, cs2)
    mr = torch.empty(B, G, 2, device=dev)
    ops.gn_fold(cs1, cs2, G, HW, 1e-5, mr)
    ws = torch.empty(ops.gn_bwd_ws_floats(B, HW, C, G), device=dev)
    oldf = nhwc(old) if acc else None
    dx1 = oldf[..., :c1].contiguous() if acc else torch.full((B, H, W, c1), float("nan"), device=dev)
    dx2 = None
    if c2:
        dx2 = oldf[..., c1:].contiguous() if acc else torch.full((B, H, W, c2), float("nan"), device=dev)
    hi = torch.empty(B, H, W, C, dtype=torch.int16, device=dev); lo = torch.empty_like(hi)
    dgam = torch.full((C,), float("nan"), device=dev); dbet = torch.full((C,), float("nan"), device=dev)
    ops.gn_bwd(x1, x2, mr, gamma.detach().float().to(dev), beta.detach().float().to(dev), G, act, nhwc(dA), nhwc(add) if use_add else None, ws,
               dx1, acc, dx2, acc, (hi, lo) if not acc else None, prec, dgam, dbet, False)
    torch.cuda.synchronize()
    got = torch.cat([dx1] + ([dx2] if c2 else []), dim=-1).double().cpu().permute(0, 3, 1, 2)
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2e-5 * scale
    assert float((dgam.double().cpu() - gamma.grad).abs().max()) <= 2e-5 * float(gamma.grad.abs().max()) + 1e-4
    assert float((dbet.double().cpu() - beta.grad).abs().max()) <= 2e-5 * float(beta.grad.abs().max()) + 1e-4
    if not acc:     # hi + lo planes (fp16 split) carry dx to ~2^-22
        rec = (hi.view(torch.float16).double() + lo.view(torch.float16).double()).cpu().permute(0, 3, 1, 2)
        assert float((rec - want).abs().max()) <= 1e-4 * scale
