#!/usr/bin/env python3
"""Phase timeline of the register-streamed kernel on the set-ViT's flat GEMMs (M = 64 x 4098 rows; run with STEDM_CONV_DBG=1024):
per-block stamps of entry / tables / loop end / tile staged / stores issued / stores retired. Timing experiment only."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("STEDM_CONV_DBG", "1024")
import numpy as np, torch
from stedm_amd import ops
from stedm_amd._lib import lib, check

def main():
    prec = ops.Precision.parse("bf16"); dev = torch.device("cuda:0")
    M = 64 * 4098
    for name, K, N, form in (("to_qkv 256->2304 (16-bit out)", 256, 2304, "o16"), ("ff1 256->256 GELU (16-bit out)", 256, 256, "o16g"),
                             ("ff2 256->256 + res (fp32 out)", 256, 256, "res"), ("to_out 768->256 + res (fp32 out)", 768, 256, "res")):
        x16 = torch.randn(M, K, device=dev).bfloat16().view(torch.int16)
        w = torch.randn(N, K, 1, 1, device=dev) / K ** 0.5
        hi, lo = ops.pack_conv_weight(w, prec); wf = ops.pack_conv_weight_frag(w, prec)
        v4 = lambda t: t.view(1, 1, M, -1)
        out = torch.randn(M, N, device=dev) if form == "res" else None
        o16 = torch.empty(M, N, dtype=torch.int16, device=dev) if form != "res" else None
        kw = dict(prec=prec, ks=1, src16=(v4(x16), None), w_frag=wf, bias=torch.randn(N, device=dev), act_out=2 if form == "o16g" else 0,
                  res=None if out is None else v4(out), out16=None if o16 is None else (v4(o16), None))
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        for _ in range(3): ops.conv_igemm(None, hi, lo, None if out is None else v4(out), **kw)
        e0.record()
        for _ in range(5): ops.conv_igemm(None, hi, lo, None if out is None else v4(out), **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        nb = min(2048, ((M + 255) // 256) * ((N + 127) // 128))
        buf = np.zeros((nb, 8), dtype=np.uint64)
        check(lib().stedm_debug_conv_stamps(buf.ctypes.data_as(ctypes.c_void_p), nb), "stamps")
        t = buf.astype(np.int64)
        med = lambda a, b: float(np.median((t[:, b] - t[:, a]) / 100.0))
        tiles = ((M + 255) // 256) * ((N + 127) // 128)
        print(f"{name}: {us:.1f} us per launch (stamps on), {tiles} tiles = {tiles / 256:.1f} per CU -> {us / (tiles / 256):.2f} us per tile slot; "
              f"block medians: entry->tables {med(0, 1):.2f}, tables->loop end {med(1, 3):.2f}, loop end->staged {med(3, 5):.2f}, "
              f"staged->stores issued {med(5, 6):.2f}, issued->retired {med(6, 4):.2f}, whole block {med(0, 4):.2f} us")

if __name__ == "__main__":
    main()
