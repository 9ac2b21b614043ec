#!/usr/bin/env python3
"""Phase timeline of the sub-pixel Upsample launches of the headline step (2x2-tap kind of conv_rs_kernel): the same stamps as
tools/conv_phases.py. Needs the diagnostic library: bash tools/conv_diag.sh 0 && STEDM_HIP_LIB=tools/_ab/lib_conv_diag0.so python3 tools/conv_phases_up.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("STEDM_CONV_DBG", "1024")
import numpy as np, torch
from stedm_amd import ops
from stedm_amd._lib import lib, check, CONV_UP_SUBPIXEL

SHAPES = [("up 1024 @8->16 B128", 128, 8, 8, 1024), ("up 512 @16->32 B128", 128, 16, 16, 512)]

def main():
    prec = ops.Precision.parse(os.environ.get("PHASES_PREC", "bf16")); dev = torch.device("cuda:0")
    for name, B, H, W, c in SHAPES:
        x = torch.randn(B, H, W, c, device=dev)
        w = torch.randn(c, c, 3, 3, device=dev) / (c * 9) ** 0.5
        hi, lo = ops.pack_conv_weight_up(w, prec); wf = ops.pack_conv_weight_up_frag(w, prec)
        h16 = torch.empty(B, H, W, c, dtype=torch.int16, device=dev)
        ops.gn_apply16(x, None, h16, None, prec)
        out = torch.empty(B, 2 * H, 2 * W, c, device=dev); bias = torch.randn(c, device=dev)
        cs = torch.empty(B, 4 * ops.gn_chan_nslab(H * W), c, 2, device=dev)
        kw = dict(prec=prec, mode=CONV_UP_SUBPIXEL, src16=(h16, None), bias=bias, w_frag=wf, chan_stats=cs)
        for _ in range(int(os.environ.get("PHASES_WARM_LAUNCHES", "200"))):
            ops.conv_igemm(None, hi, lo, out, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.conv_igemm(None, hi, lo, out, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        nb = min(2048, (B * H * W // 256) * 4 * (c // 128))
        buf = np.zeros((nb, 8), dtype=np.uint64)
        check(lib().stedm_debug_conv_stamps(buf.ctypes.data_as(ctypes.c_void_p), nb), "stamps")
        t = buf[:, :5].astype(np.int64); t0 = t[:, 0].min()
        rel = (t - t0) / 100.0
        d = np.diff(rel, axis=1)
        loop = rel[:, 3] - rel[:, 1]
        stg = (buf[:, 5].astype(np.int64) - buf[:, 3].astype(np.int64)) / 100.0
        sto = (buf[:, 6].astype(np.int64) - buf[:, 5].astype(np.int64)) / 100.0
        ret = (buf[:, 4].astype(np.int64) - buf[:, 6].astype(np.int64)) / 100.0
        cyc = buf[:, 7].astype(np.int64) - buf[:, 2].astype(np.int64)
        wall = (t[:, 3] - t[:, 1]).astype(np.float64)
        ok = (cyc > 0) & (wall > 0)
        gf = 2.0 * B * H * W * 16 * c * c / 1e9
        print(f"{name}: {us:.1f} us per launch, {gf / us * 1e3:.1f} TFLOP/s executed ({gf / us * 1e3 / 2500 * 100:.1f} % of peak); blocks {nb}", flush=True)
        if ok.any():
            ghz = np.median(cyc[ok] / (wall[ok] * 10.0))
            nmfma = (c // 16) * 4 * 8
            duty = np.median(nmfma * 32 / cyc[ok])
            print(f"    in-loop clock {ghz:.3f} GHz, MFMA duty of the loop {100 * duty:.1f} % ({nmfma} MFMAs x 32 cycles / {np.median(cyc[ok]):.0f} cycles)")
        print(f"    tables {d[:,0].mean():.1f} us, loop {loop.mean():.1f} (min {loop.min():.1f} max {loop.max():.1f}), epilogue {d[:,3].mean():.1f} (max {d[:,3].max():.1f}): "
              f"staging {stg.mean():.2f}, store loop {sto.mean():.2f}, drain {ret.mean():.2f}; start skew max {rel[:,0].max():.1f}, last end {rel[:,4].max():.1f} us", flush=True)

if __name__ == "__main__":
    main()
