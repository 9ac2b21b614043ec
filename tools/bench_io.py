"""conv_in / conv_out alone on the bench shapes (first and last convolution of the U-Net: 7 -> 128 and GN + SiLU + 128 -> 4 at 32 x 32).
    python tools/bench_io.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stedm_amd import ops

dev = torch.device("cuda:0")
def tm(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B in (64, 128):
    x = torch.randn(B, 4, 32, 32, device=dev); cc = torch.randn(B, 3, 32, 32, device=dev)
    w = torch.randn(128, 7, 3, 3, device=dev) * 0.1; b = torch.randn(128, device=dev)
    out = torch.empty(B, 32, 32, 128, device=dev); cs = torch.empty(B, 16, 128, 2, device=dev)
    us = tm(lambda: ops.conv_in(x, cc, w, b, out, chan_stats=cs))
    print(f"conv_in  B={B}: {us:6.1f} us  ({(out.numel() * 4 + x.numel() * 4 + cc.numel() * 4) / us / 1e6:.2f} TB/s of algorithmic bytes)")
    h = torch.randn(B, 32, 32, 128, device=dev)
    csh = torch.empty(B, 4, 128, 2, device=dev); ops.gn_chan_stats(h, csh)
    wo = ops.conv_out_weight(torch.randn(4, 128, 3, 3, device=dev) * 0.05)
    g = torch.ones(128, device=dev); bt = torch.zeros(128, device=dev); bo = torch.zeros(4, device=dev)
    o = torch.empty(B, 4, 32, 32, device=dev)
    us = tm(lambda: ops.conv_out(h, g, bt, 1e-5, 32, wo, bo, o, csh))
    print(f"conv_out B={B}: {us:6.1f} us  ({h.numel() * 4 / us / 1e6:.2f} TB/s of algorithmic bytes)")
