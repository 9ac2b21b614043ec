#!/usr/bin/env python3
"""Experiment: does running the CFG step as TWO half-batch passes on two streams (their tile rounds drift apart, so one pass's store bursts
overlap the other's K loops) beat one full-batch pass? Timing only; two model copies (separate buffers)."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stedm_amd.utils import prng
from stedm_amd.unet import UNetModel

NS32 = dict(image_size=32, in_channels=7, model_channels=128, out_channels=4, num_res_blocks=2, attention_resolutions=[32, 16, 8], channel_mult=[1, 4, 8], num_heads=8)
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
m1 = UNetModel(precision=prec, **NS32).eval(); prng.fill_module_(m1, seed=0); m1 = m1.to(dev)
m2 = copy.deepcopy(m1)
g = torch.Generator().manual_seed(1)
def inputs(b):
    return (torch.randn(b, 4, 32, 32, generator=g).to(dev), torch.randn(b, 3, 32, 32, generator=g).to(dev), torch.full((b,), 500, dtype=torch.int64, device=dev),
            torch.randn(b, 512, generator=g).to(dev), torch.randn(b, 512, generator=g).to(dev))
def run(m, inp, out):
    x, cc, t, c1, c2 = inp
    m._forward_impl(x, cc, t, [c1, c2], out, uniform_t=True)
full = inputs(B); h1 = inputs(B // 2); h2 = inputs(B // 2)
of = torch.empty(2 * B, 4, 32, 32, device=dev); o1 = torch.empty(B, 4, 32, 32, device=dev); o2 = torch.empty(B, 4, 32, 32, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
N = 30
for _ in range(3):
    run(m1, full, of)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N):
    run(m1, full, of)
torch.cuda.synchronize(); tf = (time.perf_counter() - t0) / N
for _ in range(3):
    run(m1, h1, o1); run(m2, h2, o2)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N):
    run(m1, h1, o1); run(m2, h2, o2)
torch.cuda.synchronize(); ts = (time.perf_counter() - t0) / N
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N):
    with torch.cuda.stream(s1):
        run(m1, h1, o1)
    with torch.cuda.stream(s2):
        run(m2, h2, o2)
torch.cuda.synchronize(); tc = (time.perf_counter() - t0) / N
print(f"{prec} B={B}: one pass {tf * 1e3:.3f} ms | two half passes, one stream {ts * 1e3:.3f} ms | two half passes, two streams {tc * 1e3:.3f} ms")
