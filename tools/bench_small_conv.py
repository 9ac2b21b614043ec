"""One split-K convolution of the batch-1 CFG step (2 samples), warm against cold weights: the same layer back to back (its 16-bit weights
stay in L2 / the Infinity Cache) vs 24 layers' worth of distinct weights in rotation (every launch streams its weights from HBM, as inside a
denoising step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stedm_amd import ops

dev = torch.device("cuda:0")
prec = ops.Precision.parse("bf16")
for (B, H, cin, cout) in ((2, 8, 1024, 1024), (2, 16, 512, 512), (2, 32, 128, 128), (2, 8, 2048, 1024)):
    NW = 24
    x = torch.randn(B, H, H, cin, device=dev)
    h16 = torch.empty(B, H, H, cin, dtype=torch.int16, device=dev)
    ops.gn_apply16(x, None, h16, None, prec)
    ws_ = []
    for i in range(NW):
        w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
        ws_.append((ops.LazyPlanes(lambda w=w: ops.pack_conv_weight(w, prec)), ops.pack_conv_weight_frag(w, prec),
                    ops.pack_conv_weight_frag16(w, prec) if cin >= 256 else None))
    out = torch.empty(B, H, H, cout, device=dev)
    bias = torch.randn(cout, device=dev)
    wsp = torch.empty(16 * out.numel(), device=dev)
    def run(i):
        hi, fr, fr16 = ws_[i]
        ops.conv_igemm(None, hi, None, out, prec=prec, ks=3, src16=(h16, None), bias=bias, w_frag=fr, w_frag16=fr16, ws=wsp)
    for mode, idx in (("warm", lambda k: 0), ("cold", lambda k: k % NW)):
        for k in range(NW): run(idx(k))
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        n = 96
        e0.record()
        for k in range(n): run(idx(k))
        e1.record(); torch.cuda.synchronize()
        print(f"3x3 {cin}->{cout} @{H}x{H} B={B} {mode}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per conv (+ reduce)")
