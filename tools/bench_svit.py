#!/usr/bin/env python3
"""Throughput of the set-ViT style encoder (sViT, 4 style images of 512x512, T = 4098 tokens, dim 256, 12 heads,
depth 6) and the MFMA rate of its flash attention kernel (309.5 GFLOP per sample). Synthetic inputs, PRNG weights."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stedm_amd.style import sViT
from stedm_amd.utils import prng

def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    m = sViT(image_size=512, patch_size=8, num_classes=512, dim=256, depth=6, heads=12, mlp_dim=256, pool="mean", channels=3,
             dim_head=64, dropout=0.0, emb_dropout=0.0, ns=4, t_dim=256, precision=prec).eval()
    prng.fill_module_(m, seed=7)
    m = m.to(dev)
    x = (torch.rand(B, 4, 512, 512, 3, device=dev) * 2 - 1)
    for _ in range(2):
        y = m(x)
    torch.cuda.synchronize()
    n = 5
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        y = m(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    gf = 356.3 * B     # GFLOP per sample (46.7 GEMM + 309.5 attention), SURVEY §8d
    print(f"sViT {prec} B={B}: {ms:.2f} ms per forward, {B / ms * 1e3:.1f} samples/s, {gf / ms:.1f} TFLOP/s overall "
          f"({gf / ms / 2500 * 100:.1f} % of the 2.5 PF dense peak); finite: {bool(torch.isfinite(y).all())}")

if __name__ == "__main__":
    main()
