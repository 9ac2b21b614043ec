#!/usr/bin/env python3
"""Does an HBM-bound gn_apply16c launch overlap an MFMA-bound conv_rs launch issued on another stream? (timing experiment)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stedm_amd import ops

def main():
    dev = torch.device("cuda:0"); prec = ops.Precision.parse("bf16")
    B, H, W, cin, cout = 64, 16, 16, 512, 512
    x = torch.randn(B, H, W, cin, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
    hi, lo = ops.pack_conv_weight(w, prec); wf = ops.pack_conv_weight_frag(w, prec)
    h16 = torch.empty(B, H, W, cin, dtype=torch.int16, device=dev)
    ops.gn_apply16(x, None, h16, None, prec)
    out = torch.empty(B, H, W, cout, device=dev)
    # an independent GroupNorm apply of the same size class as the decoder's
    y = torch.randn(B, 32, 32, 256, device=dev)
    cs = torch.empty(B, 4, 256, 2, device=dev); ops.gn_chan_stats(y, cs)
    g = torch.ones(256, device=dev); bt = torch.zeros(256, device=dev)
    y16 = torch.empty(B, 32, 32, 256, dtype=torch.int16, device=dev)
    conv = lambda: ops.conv_igemm(None, hi, lo, out, prec=prec, src16=(h16, None), w_frag=wf)
    app = lambda: ops.gn_apply16c(y, cs, None, None, y16, None, prec, g, bt, 1e-5, 32, 1)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def timed(fn, n=20):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(n); e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    def seq(n):
        for _ in range(n): conv(); app()
    def par(n):
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        for _ in range(n):
            with torch.cuda.stream(s1): conv()
            with torch.cuda.stream(s2): app()
        cur.wait_stream(s1); cur.wait_stream(s2)
    for f in (lambda n: [conv() for _ in range(n)], lambda n: [app() for _ in range(n)], seq, par): f(3)
    print(f"conv alone {timed(lambda n: [conv() for _ in range(n)]):.1f} us, apply alone {timed(lambda n: [app() for _ in range(n)]):.1f} us, "
          f"sequential {timed(seq):.1f} us, two streams {timed(par):.1f} us per pair")

if __name__ == "__main__":
    main()
