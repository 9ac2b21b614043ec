#!/usr/bin/env python3
"""Micro-benchmark of stedm_conv_igemm on the NS32 layer shapes (B=64 encoder / B=128 CFG decoder).
Prints TFLOP/s per shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stedm_amd import ops
from stedm_amd._lib import CONV_S1, CONV_UP, CONV_DOWN, CONV_UP_SUBPIXEL

SHAPES = [  # name, B, H, W, c1, c2, cout, mode, ks, gn
    ("L0 128->128 @32 B64", 64, 32, 32, 128, 0, 128, CONV_S1, 3, True),
    ("L0 256->128 @32 B128", 128, 32, 32, 128, 128, 128, CONV_S1, 3, True),
    ("L0 640->128 @32 B128", 128, 32, 32, 512, 128, 128, CONV_S1, 3, True),
    ("L1 512->512 @16 B64", 64, 16, 16, 512, 0, 512, CONV_S1, 3, True),
    ("L1 512->512 @16 B128", 128, 16, 16, 512, 0, 512, CONV_S1, 3, True),
    ("L1 1536->512 @16 B128", 128, 16, 16, 1024, 512, 512, CONV_S1, 3, True),
    ("L2 1024->1024 @8 B64", 64, 8, 8, 1024, 0, 1024, CONV_S1, 3, True),
    ("L2 1024->1024 @8 B128", 128, 8, 8, 1024, 0, 1024, CONV_S1, 3, True),
    ("L2 2048->1024 @8 B128", 128, 8, 8, 1024, 1024, 1024, CONV_S1, 3, True),
    ("UP 1024 8->16 B128", 128, 8, 8, 1024, 0, 1024, CONV_UP, 3, False),
    ("UP 512 16->32 B128", 128, 16, 16, 512, 0, 512, CONV_UP, 3, False),
    ("UPSUB 1024 8->16 B128", 128, 8, 8, 1024, 0, 1024, CONV_UP_SUBPIXEL, 3, False),
    ("UPSUB 512 16->32 B128", 128, 16, 16, 512, 0, 512, CONV_UP_SUBPIXEL, 3, False),
    ("DOWN 512 16->8 B64", 64, 16, 16, 512, 0, 512, CONV_DOWN, 3, False),
    ("1x1 2048->1024 @8 B128", 128, 8, 8, 1024, 1024, 1024, CONV_S1, 1, False),
    ("1x1 qkv 1024->3072 @8 B128", 128, 8, 8, 1024, 0, 3072, CONV_S1, 1, True),
]

def main():
    prec = ops.Precision.parse(sys.argv[1] if len(sys.argv) > 1 else "bf16")
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    tot_f = tot_t = 0.0
    flt = os.environ.get('BENCH_FILTER')
    for name, B, H, W, c1, c2, cout, mode, ks, gn in SHAPES:
        if flt and not any(f in name for f in flt.split('|')): continue
        cin = c1 + c2
        x1 = torch.randn(B, H, W, c1, device=dev)
        x2 = torch.randn(B, H, W, c2, device=dev) if c2 else None
        w = torch.randn(cout, cin, ks, ks, device=dev) / (cin * ks * ks) ** 0.5
        hi, lo = ops.pack_conv_weight_up(w, prec) if mode == CONV_UP_SUBPIXEL else ops.pack_conv_weight(w, prec)
        sc = torch.rand(B, cin, device=dev) + 0.5 if gn else None
        sh = torch.randn(B, cin, device=dev) * 0.1 if gn else None
        Ho, Wo = (H * 2, W * 2) if mode in (CONV_UP, CONV_UP_SUBPIXEL) else ((H // 2, W // 2) if mode == CONV_DOWN else (H, W))
        out = torch.empty(B, Ho, Wo, cout, device=dev)
        bias = torch.randn(cout, device=dev)
        if os.environ.get("BENCH_DMA"):
            C = cin
            h16 = torch.empty(B, H, W, C, dtype=torch.int16, device=dev); l16 = torch.empty_like(h16)
            ops.gn_apply16(x1, x2, h16, l16 if prec.npass == 3 else None, prec)
            wf = None
            if os.environ.get("BENCH_FRAG") and prec.npass == 1 and mode in (CONV_S1, CONV_UP_SUBPIXEL):
                wf = ops.pack_conv_weight_up_frag(w, prec) if mode == CONV_UP_SUBPIXEL else ops.pack_conv_weight_frag(w, prec)
            run = lambda: ops.conv_igemm(None, hi, lo, out, prec=prec, ks=ks, mode=mode, src16=(h16, l16), bias=bias, w_frag=wf, ws=torch.empty(2 * out.numel(), device=dev) if os.environ.get('BENCH_WS') else None)
        else:
            run = lambda: ops.conv_igemm(x1, hi, lo, out, prec=prec, ks=ks, mode=mode, src2=x2, scale=sc, shift=sh, act=1 if gn else 0, bias=bias)
        for _ in range(3): run()
        torch.cuda.synchronize()
        n = 10
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        fl = 2.0 * B * Ho * Wo * cout * cin * (4 if mode == CONV_UP_SUBPIXEL else ks * ks)
        tot_f += fl; tot_t += ms
        print(f"{name:32s} {ms*1e3:9.1f} us  {fl/ms/1e9:8.1f} TFLOP/s  ({fl/ms/1e9/2500*100:5.1f}% of peak)", flush=True)
    print(f"{'SUM':32s} {tot_t*1e3:9.1f} us  {tot_f/tot_t/1e9:8.1f} TFLOP/s")

if __name__ == "__main__":
    main()
