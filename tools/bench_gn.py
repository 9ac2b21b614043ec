#!/usr/bin/env python3
"""gn_apply16c alone: GroupNorm + SiLU from producer-side channel partials -> 16-bit planes, on the shapes of the denoising step.
    python tools/bench_gn.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stedm_amd import ops
from stedm_amd.ops import Precision

dev = torch.device("cuda:0")
prec = Precision.parse("bf16")
SHAPES = [("dec 16^2 1024+512 B128", 128, 16, 1024, 512), ("dec 8^2 1024+1024 B128", 128, 8, 1024, 1024), ("dec 32^2 128+128 B128", 128, 32, 128, 128),
          ("enc 16^2 512 B64", 64, 16, 512, 0), ("dec 16^2 512 B128", 128, 16, 512, 0), ("big 16^2 1024+512 B512", 512, 16, 1024, 512)]
for name, B, H, c1, c2 in SHAPES:
    x1 = torch.randn(B, H, H, c1, device=dev)
    x2 = torch.randn(B, H, H, c2, device=dev) if c2 else None
    ns = ops.gn_chan_nslab(H * H)
    cs1 = torch.empty(B, ns, c1, 2, device=dev); ops.gn_chan_stats(x1, cs1)
    cs2 = None
    if c2:
        cs2 = torch.empty(B, ns, c2, 2, device=dev); ops.gn_chan_stats(x2, cs2)
    C = c1 + c2
    hi = torch.empty(B, H, H, C, dtype=torch.int16, device=dev)
    g = torch.ones(C, device=dev); bt = torch.zeros(C, device=dev)
    run = lambda: ops.gn_apply16c(x1, cs1, x2, cs2, hi, None, prec, g, bt, 1e-5, 32, 1)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    mb = B * H * H * C * 6 / 1e6
    print(f"{name:28s} {us:7.1f} us  {mb:7.1f} MB  {mb / us:6.2f} TB/s")
