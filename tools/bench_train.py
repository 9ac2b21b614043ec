"""Training-step timing (BASELINE config 2: 32x32x4 latents, batch 64, bf16): forward + L1 + backward + AdamW/EMA."""
import argparse, sys, time
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stedm_amd.utils import prng
from stedm_amd.unet import UNetModel
from stedm_amd.train import UNetTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--parts", action="store_true")
ap.add_argument("--latent", type=int, default=32)
ap.add_argument("--graph", action="store_true", help="UNetTrainer.train_step_graphed: the step captured once and replayed as one hipGraph launch")
a = ap.parse_args()
NS32 = dict(image_size=32, in_channels=7, model_channels=128, out_channels=4, num_res_blocks=2, attention_resolutions=[32, 16, 8], channel_mult=[1, 4, 8], num_heads=8)
dev = torch.device("cuda:0")
m = UNetModel(precision=a.precision, **NS32).eval(); prng.fill_module_(m, seed=0); m = m.to(dev)
tr = UNetTrainer(m, lr=1e-5)
B = a.batch
g = torch.Generator(device="cpu").manual_seed(1)
L = a.latent
x = torch.randn(B, 4, L, L, generator=g).to(dev); cc = torch.randn(B, 3, L, L, generator=g).to(dev)
ctx = torch.randn(B, 512, generator=g).to(dev); tgt = torch.randn(B, 4, L, L, generator=g).to(dev)
t = torch.randint(0, 1000, (B,), generator=g).to(dev)
step = tr.train_step_graphed if a.graph else tr.train_step
for _ in range(a.warmup + (tr.GRAPH_WARMUP + 1 if a.graph else 0)):
    loss = step(x, cc, t, ctx, tgt)
torch.cuda.synchronize()
if a.graph:
    assert tr._graph is not None
t0 = time.perf_counter()
for _ in range(a.steps):
    loss = step(x, cc, t, ctx, tgt)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(f"train step{' (hipGraph replay)' if a.graph else ''} B={B} latent {L}x{L} {a.precision}: {dt * 1e3:.2f} ms  ({1 / dt:.2f} steps/s, {B / dt:.0f} samples/s)  loss {float(loss):.4f}")
# host side of the same step: time until train_step returns (launches issued, nothing awaited) and the GPU time between two events
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
h = []
e0.record()
for _ in range(a.steps):
    t0 = time.perf_counter(); step(x, cc, t, ctx, tgt); h.append(time.perf_counter() - t0)
e1.record(); torch.cuda.synchronize()
print(f"host issue time per step {1e3 * sum(h) / len(h):.2f} ms (min {1e3 * min(h):.2f}); stream time per step {e0.elapsed_time(e1) / a.steps:.2f} ms")
if a.parts:
    def tm(f, n=3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    print("forward   %.2f ms" % tm(lambda: tr.forward(x, cc, t, ctx)))
    dp = torch.randn(B, 4, L, L, device=dev) * 1e-5
    print("backward  %.2f ms" % tm(lambda: tr.backward(dp)))
    tr._grads_ready = True
    def opt():
        tr._grads_ready = True; tr.optimizer_step()
    print("optimizer %.2f ms" % tm(opt))
    print("repack    %.2f ms" % tm(lambda: (m.invalidate(), m._prepare())))
