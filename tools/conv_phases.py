#!/usr/bin/env python3
"""Phase timeline of the register-streamed 3x3 conv kernel (run with STEDM_CONV_DBG=1024): per-block stamps of
entry / tables / first patch / loop end / stores retired, summarised over the grid. Timing experiment only."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("STEDM_CONV_DBG", "1024")
import numpy as np, torch
from stedm_amd import ops
from stedm_amd._lib import lib, check

SHAPES = [("L0 128->128 @32 B64", 64, 32, 32, 128, 128), ("L0 640->128 @32 B128", 128, 32, 32, 640, 128),
          ("L1 512->512 @16 B128", 128, 16, 16, 512, 512), ("L1 1536->512 @16 B128", 128, 16, 16, 1536, 512),
          ("L2 2048->1024 @8 B128", 128, 8, 8, 2048, 1024)]

SMALL = [("L2 1024->1024 @8 B2", 2, 8, 8, 1024, 1024), ("L1 512->512 @16 B2", 2, 16, 16, 512, 512), ("L0 128->128 @32 B2", 2, 32, 32, 128, 128),
         ("L2 2048->1024 @8 B2", 2, 8, 8, 2048, 1024)]      # PHASES_SMALL=1: the split-K launches of the batch-1 CFG step

def main():
    prec = ops.Precision.parse("bf16"); dev = torch.device("cuda:0")
    small = bool(os.environ.get("PHASES_SMALL"))
    for name, B, H, W, cin, cout in (SMALL if small else SHAPES):
        if os.environ.get("PHASES_B"):       # fewer tiles than CUs: is the epilogue's store burst bound per CU or by the fabric / HBM?
            B = int(os.environ["PHASES_B"]); name += f" (B -> {B})"
        x = torch.randn(B, H, W, cin, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
        hi, lo = ops.pack_conv_weight(w, prec); wf = ops.pack_conv_weight_frag(w, prec)
        wf16 = ops.pack_conv_weight_frag16(w, prec) if cin >= 256 and not os.environ.get("PHASES_NO_M16") else None   # the 16x16x32 kind (the default from 256 channels)
        h16 = torch.empty(B, H, W, cin, dtype=torch.int16, device=dev)
        ops.gn_apply16(x, None, h16, None, prec)
        if os.environ.get("PHASES_ZERO_A"): h16.zero_()        # data-dependence experiment: all-zero activations
        if os.environ.get("PHASES_ZERO_W"): wf.zero_()
        out = torch.empty(B, H, W, cout, device=dev); bias = torch.randn(cout, device=dev)
        cs = torch.empty(B, (H * W + 255) // 256, cout, 2, device=dev)
        nwarm = int(os.environ.get("PHASES_WARM_LAUNCHES", "3"))       # in-loop clock: >= 2 s of back-to-back launches before the stamped one
        for _ in range(nwarm):
            ops.conv_igemm(None, hi, lo, out, prec=prec, src16=(h16, None), bias=bias, w_frag=wf, res=(out if os.environ.get("PHASES_RES") else None),
                           chan_stats=(cs if os.environ.get("PHASES_STATS") else None), w_frag16=wf16,
                           ws=(torch.empty(16 * out.numel(), device=dev) if small else None))
        torch.cuda.synchronize()
        nb = min(2048, max(1, B * H * W // 256) * ((cout + 127) // 128) * (16 if small else 1))
        buf = np.zeros((nb, 8), dtype=np.uint64)
        check(lib().stedm_debug_conv_stamps(buf.ctypes.data_as(ctypes.c_void_p), nb), "stamps")
        t = buf[:, :5].astype(np.int64); t0 = t[:, 0].min()
        rel = (t - t0) / 100.0   # us
        d = np.diff(rel, axis=1)
        loop = rel[:, 3] - rel[:, 1]    # tables built -> main loop done (first patch wait included)
        stg = (buf[:, 5].astype(np.int64) - buf[:, 3].astype(np.int64)) / 100.0      # loop done -> tile staged in LDS (barriers #3, #4)
        sto = (buf[:, 6].astype(np.int64) - buf[:, 5].astype(np.int64)) / 100.0      # staged -> store loop issued (incl. statistics)
        ret = (buf[:, 4].astype(np.int64) - buf[:, 6].astype(np.int64)) / 100.0      # issued -> stores retired (vmcnt(0); not paid by a real run)
        # in-loop shader clock and MFMA duty of compute wave 0 of every block: slots 2 / 7 are s_memtime at loop start / end, slots 1 / 3 the
        # 100 MHz wall clock at the same points
        cyc = buf[:, 7].astype(np.int64) - buf[:, 2].astype(np.int64)
        wall = (t[:, 3] - t[:, 1]).astype(np.float64)                  # 10 ns ticks
        ok = (cyc > 0) & (wall > 0)
        if ok.any():
            ghz = np.median(cyc[ok] / (wall[ok] * 10.0))               # cycles per ns
            m16 = wf16 is not None
            nch = cin // 32 if m16 else cin // 16
            nmfma = nch * 9 * (32 if m16 else 8) // (16 if small else 1)
            duty = np.median(nmfma * (16 if m16 else 32) / cyc[ok])
            print(f"    in-loop clock {ghz:.3f} GHz (median over blocks; delta s_memtime / delta s_memrealtime), MFMA duty of the loop {100 * duty:.1f} % "
                  f"({nmfma} MFMAs x {16 if m16 else 32} cycles per compute wave / {np.median(cyc[ok]):.0f} cycles); loop rate = "
                  f"{2.0 * 256 * 128 * cin * 9 / (np.median(wall[ok]) * 1e-8) / 1e12 * 256 / 1e3:.3f} PFLOP/s if all 256 CUs ran this loop")
        print(f"    epilogue split: staging {stg.mean():.2f} us, store loop {sto.mean():.2f} us, drain {ret.mean():.2f} us")
        print(f"{name}: blocks {nb}  start skew max {rel[:,0].max():.1f} us | tables {d[:,0].mean():.1f}  loop {loop.mean():.1f} (min {loop.min():.1f} "
              f"max {loop.max():.1f})  epilogue {d[:,3].mean():.1f} (max {d[:,3].max():.1f}) | last end {rel[:,4].max():.1f} us", flush=True)

if __name__ == "__main__":
    main()
