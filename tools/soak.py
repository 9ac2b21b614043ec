"""Sustained-load check: the same DDIM-50 + CFG sampling run (B = 64) repeated N times must give bit-identical latents every time (graph replay,
fused epilogues, split-K reduces: any race or uninitialised read shows as a difference), then M training steps on fresh batches must keep the
loss finite and falling. Prints one line per phase; exits non-zero on a failure."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stedm_amd.utils import prng
from stedm_amd.unet import UNetModel
from stedm_amd.train import UNetTrainer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
M = int(sys.argv[2]) if len(sys.argv) > 2 else 150
dev = torch.device("cuda:0")
for precision in ("bf16", "f16"):
    ld = bench.build_model(dev, precision)
    xT, cond, unc = bench.synth_inputs(dev, 64, 0)
    ref = None
    t0 = time.perf_counter()
    for i in range(N):
        with torch.no_grad():
            z, _ = ld.sample_log(cond, 64, True, 50, eta=0.0, unconditional_conditioning=unc, unconditional_guidance_scale=1.5, x_T=xT.clone())
        torch.cuda.synchronize()
        assert bool(torch.isfinite(z).all()), f"{precision}: non-finite latents in run {i}"
        if ref is None:
            ref = z.clone()
        else:
            assert torch.equal(z, ref), f"{precision}: run {i} differs from run 0 (max |diff| {float((z - ref).abs().max()):.3e})"
    dt = time.perf_counter() - t0
    print(f"sampling {precision}: {N} x DDIM-50 at B = 64 bit-identical, {N * 50 / dt:.1f} steps/s including the style-free loop overheads", flush=True)
    del ld
NS32 = bench.NS32
m = UNetModel(precision="bf16", **NS32).eval(); prng.fill_module_(m, seed=0); m = m.to(dev)
tr = UNetTrainer(m, lr=1e-4, weight_decay=0.01)
g = torch.Generator(device="cpu").manual_seed(1)
losses = []
t0 = time.perf_counter()
for it in range(M):
    cc = (torch.randn(64, 3, 32, 32, generator=g) > 0).float().to(dev)
    ctx = torch.randn(64, 512, generator=g).to(dev); noise = torch.randn(64, 4, 32, 32, generator=g).to(dev)
    t = torch.randint(0, 1000, (64,), generator=g).to(dev)
    losses.append(float(tr.train_step(noise, cc, t, ctx, noise)))
dt = time.perf_counter() - t0
assert all(v == v and v < 10 for v in losses), "non-finite or exploding loss"
assert sum(losses[-10:]) / 10 < losses[0] - 0.05, "the loss did not fall"
assert all(bool(torch.isfinite(p).all()) for p in m.parameters())
print(f"training bf16: {M} steps, loss {losses[0]:.4f} -> {sum(losses[-10:]) / 10:.4f} (mean of the last 10), {dt / M * 1e3:.2f} ms per step incl. host batch generation", flush=True)
