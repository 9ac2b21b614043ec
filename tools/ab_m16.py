#!/usr/bin/env python3
"""Same-box, same-process A/B of the two MFMA shapes of the register-streamed 3x3 kernel (conv_rs.inc): RS_3X3 on
v_mfma_f32_32x32x16 vs RS_3X3M on v_mfma_f32_16x16x32, interleaved rounds on random data (cdna_hip_programming.md §5.4 rules 24, 25,
28), per layer shape of the NS32 denoising step and for the whole hipGraph-replayed step.

    python tools/ab_m16.py [bf16|f16] [rounds]
    python tools/ab_m16.py parity [rounds]     3-product mode: LDS-operand kernel vs the register-streamed 16x16x32 kind (conv_rs.inc P3)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stedm_amd import ops

SHAPES = [  # name, B, H, W, cin, cout (plain 3x3 stride 1 of the CFG step: encoder at B=64, decoder at B=128)
    ("128->128 @32 B64", 64, 32, 32, 128, 128), ("128->128 @32 B128", 128, 32, 32, 128, 128), ("256->128 @32 B128", 128, 32, 32, 256, 128),
    ("512->512 @16 B64", 64, 16, 16, 512, 512), ("512->512 @16 B128", 128, 16, 16, 512, 512), ("1536->512 @16 B128", 128, 16, 16, 1536, 512),
    ("1024->1024 @8 B64", 64, 8, 8, 1024, 1024), ("1024->1024 @8 B128", 128, 8, 8, 1024, 1024), ("2048->1024 @8 B128", 128, 8, 8, 2048, 1024),
]


def main():
    prec = ops.Precision.parse(sys.argv[1] if len(sys.argv) > 1 else "bf16")
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    print(f"{'shape':24s} {'32x32x16 us':>12s} {'16x16x32 us':>12s} {'ratio':>7s} {'TF/s 32':>9s} {'TF/s 16':>9s}")
    tot = [0.0, 0.0]
    for name, B, H, W, cin, cout in SHAPES:
        x = torch.randn(B, H, W, cin, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
        h16 = torch.empty(B, H, W, cin, dtype=torch.int16, device=dev)
        l16 = torch.empty_like(h16) if prec.npass == 3 else None
        ops.gn_apply16(x, None, h16, l16, prec)
        hi, lo = ops.pack_conv_weight(w, prec)
        wf, wf16 = (ops.pack_conv_weight_frag(w, prec) if prec.npass == 1 else None), ops.pack_conv_weight_frag16(w, prec)
        out = torch.empty(B, H, W, cout, device=dev)
        bias = torch.randn(cout, device=dev)
        ws = torch.empty(16 * out.numel(), device=dev) if out.numel() <= (1 << 20) else torch.empty(2 * out.numel(), device=dev)
        cs = torch.empty(B, (H * W + 255) // 256, cout, 2, device=dev)
        runs = [lambda f16=f16: ops.conv_igemm(None, hi, lo, out, prec=prec, src16=(h16, l16), bias=bias, w_frag=wf, w_frag16=f16, ws=ws, chan_stats=cs)
                for f16 in (None, wf16)]
        for r in runs:
            for _ in range(3): r()
        torch.cuda.synchronize()
        ts = [[], []]
        for _ in range(rounds):
            for k, r in enumerate(runs):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): r()
                e1.record(); torch.cuda.synchronize()
                ts[k].append(e0.elapsed_time(e1) / 5 * 1e3)
        m = [float(np.median(t)) for t in ts]
        fl = 2.0 * B * H * W * cout * cin * 9 * prec.npass      # executed products
        tot[0] += m[0]; tot[1] += m[1]
        print(f"{name:24s} {m[0]:12.1f} {m[1]:12.1f} {m[0] / m[1]:7.3f} {fl / m[0] / 1e6:9.1f} {fl / m[1] / 1e6:9.1f}", flush=True)
    print(f"{'SUM':24s} {tot[0]:12.1f} {tot[1]:12.1f} {tot[0] / tot[1]:7.3f}")

    # whole denoising step (hipGraph replay), the two kinds interleaved
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    ld = bench.build_model(dev, prec.label if prec.npass == 1 else "parity")
    unet = ld.model.diffusion_model
    xT, cond, unc = bench.synth_inputs(dev, 64, 0)
    res = {0: [], 1: []}
    for rd in range(4):
        for k in (0, 1):
            unet._m16 = bool(k)
            unet.invalidate()
            dt, _ = bench.run_steps(ld, xT, cond, unc, 3, 20, 1)
            res[k].append(dt / 20 * 1e3)
    a, b = float(np.median(res[0])), float(np.median(res[1]))
    print(f"denoising step (B=64 CFG, graph replay): 32x32x16 {a:.3f} ms = {1e3 / a:.1f} steps/s | 16x16x32 on the plain 3x3 convs {b:.3f} ms = {1e3 / b:.1f} steps/s | x{a / b:.3f}")


if __name__ == "__main__":
    main()
