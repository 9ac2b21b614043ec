#!/usr/bin/env python3
"""The U-Net / SpatialTransformer attention kernel alone (stedm_attn_legacy16 -> attn_flash_kernel) on the 16-bit qkv plane.
    python tools/bench_attn.py [prec] [B] [T] [heads] [ch]        (defaults: f16 32 1024 8 128 = one REF128 CFG pass at batch 16)
    STEDM_ATTN_TILES=1: round 4's attn_mfma_tiles_kernel (A/B)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stedm_amd import ops

prec = ops.Precision.parse(sys.argv[1] if len(sys.argv) > 1 else "f16")
B, T, H, ch = [int(v) for v in (sys.argv[2:6] + ["32", "1024", "8", "128"][len(sys.argv[2:6]):])]
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(3)
qkv = (torch.randn(B, T, 3 * H * ch, generator=g) * 1.5).to(dev)
q16 = torch.empty(qkv.shape, dtype=torch.int16, device=dev)
ops.gn_apply16(qkv.view(B, 1, T, -1), None, q16.view(B, 1, T, -1), None, prec)
out = torch.empty((B, T, H * ch), dtype=torch.int16, device=dev)
for _ in range(5):
    ops.attn_legacy16(q16, out, H, prec)
torch.cuda.synchronize()
n = int(os.environ.get("ATTN_ITERS", "50"))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    ops.attn_legacy16(q16, out, H, prec)
e1.record(); torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / n
fl = 4.0 * B * H * T * T * ch
print(json.dumps({"kernel": "attn_mfma_tiles (r04)" if os.environ.get("STEDM_ATTN_TILES") else "attn_flash", "prec": prec.label, "B": B, "T": T, "heads": H,
                  "ch": ch, "us": round(us, 1), "tflops": round(fl / us / 1e6, 1), "frac_of_2500": round(fl / us / 1e6 / 2500.0, 4)}), flush=True)
