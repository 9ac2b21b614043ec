#!/usr/bin/env python3
"""Throughput of the Swin-V2-T style embedder (Agg_Mean / Agg_Max / Agg_Linear run it on (B n) x 512 x 512 style images).
Synthetic inputs, trunc-normal weights.   python tools/bench_swin.py [bf16|f16|parity] [images] [chunk]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stedm_amd.swin import swin_v2_t


def gflop_per_image(H=512, W=512, embed=96, depths=(2, 2, 6, 2)):
    h, w, dim, tot = H // 4, W // 4, embed, 0.0
    tot += 2 * h * w * 48 * embed
    for s, d in enumerate(depths):
        T = h * w
        per_block = 2 * T * dim * (3 * dim + dim + 8 * dim) + 2 * T * 64 * 2 * dim     # qkv, proj, mlp; window attention (QK^T and PV over 64 keys)
        tot += d * per_block
        if s < len(depths) - 1:
            h, w = h // 2, w // 2
            tot += 2 * h * w * 4 * dim * 2 * dim
            dim *= 2
    return tot / 1e9


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    m = swin_v2_t(num_classes=512, precision=prec, chunk_images=chunk).eval().to(dev)
    x = (torch.rand(N, 512, 512, 3, device=dev) * 2 - 1).permute(0, 3, 1, 2)      # the '(b n) c h w' view of NHWC images
    for _ in range(2):
        y = m(x)
    torch.cuda.synchronize()
    n = 3
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        y = m(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    gf = gflop_per_image() * N
    print(f"swin_v2_t {prec} N={N} chunk={chunk}: {ms:.2f} ms per forward, {N / ms * 1e3:.1f} images/s, {gf / ms:.1f} TFLOP/s "
          f"({gflop_per_image():.1f} GFLOP per image); finite: {bool(torch.isfinite(y).all())}")


if __name__ == "__main__":
    main()
