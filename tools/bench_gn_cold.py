#!/usr/bin/env python3
"""gn_apply16c with cold operands: NSETS buffer sets in rotation (more bytes than the 256 MB Infinity Cache), single-source against concat
inputs of the same total width — is the concat form slower by itself, or because its skip operand comes from HBM inside a step?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stedm_amd import ops
from stedm_amd.ops import Precision
dev = torch.device("cuda:0"); prec = Precision.parse("bf16")
NSETS = 12
for name, B, H, c1, c2 in (("single 256", 128, 32, 256, 0), ("concat 128+128", 128, 32, 128, 128), ("single 640", 128, 32, 640, 0), ("concat 512+128", 128, 32, 512, 128),
                            ("single 1536 @16", 128, 16, 1536, 0), ("concat 1024+512 @16", 128, 16, 1024, 512)):
    sets = []
    for i in range(NSETS):
        x1 = torch.randn(B, H, H, c1, device=dev); x2 = torch.randn(B, H, H, c2, device=dev) if c2 else None
        ns = ops.gn_chan_nslab(H * H)
        cs1 = torch.empty(B, ns, c1, 2, device=dev); ops.gn_chan_stats(x1, cs1)
        cs2 = None
        if c2:
            cs2 = torch.empty(B, ns, c2, 2, device=dev); ops.gn_chan_stats(x2, cs2)
        hi = torch.empty(B, H, H, c1 + c2, dtype=torch.int16, device=dev)
        sets.append((x1, cs1, x2, cs2, hi))
    C = c1 + c2
    g = torch.ones(C, device=dev); bt = torch.zeros(C, device=dev)
    def run(i):
        x1, cs1, x2, cs2, hi = sets[i % NSETS]
        ops.gn_apply16c(x1, cs1, x2, cs2, hi, None, prec, g, bt, 1e-5, 32, 1)
    for mode, f in (("cold", lambda k: k), ("warm", lambda k: 0)):
        for k in range(NSETS): run(f(k))
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        n = 48
        e0.record()
        for k in range(n): run(f(k))
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        mb = B * H * H * C * 6 / 1e6
        print(f"{name:22s} {mode}: {us:7.1f} us  {mb:7.1f} MB  {mb / us:5.2f} TB/s")
    del sets
