import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stedm_amd import ops
dev = torch.device("cuda:0")
pr = ops.Precision.parse("f16")
def run(B, H, W, cin, cout, use_ws, use_cs, use_res):
    torch.manual_seed(0)
    x = torch.randn(B, H, W, cin, device=dev)
    w = torch.randn(cout, cin, 1, 1, device=dev) / math.sqrt(cin)
    bias = torch.randn(cout, device=dev); res = torch.randn(B, H, W, cout, device=dev)
    hi16 = torch.empty((B, H, W, cin), dtype=torch.int16, device=dev)
    ops.gn_apply16(x, None, hi16, None, pr)
    whi, wlo = ops.pack_conv_weight(w, pr)
    out = torch.full((B, H, W, cout), float("nan"), device=dev)
    cs = torch.full((B, ops.gn_chan_nslab(H * W), cout, 2), float("nan"), device=dev)
    ops.conv_igemm(None, whi, wlo, out, prec=pr, ks=1, src16=(hi16, None), bias=bias, res=res if use_res else None,
                   w_frag=ops.pack_conv_weight_frag(w, pr), chan_stats=cs if use_cs else None, ws=torch.empty(16 * out.numel(), device=dev) if use_ws else None)
    xr = hi16.view(torch.float16).double()
    ref = torch.einsum("bhwc,oc->bhwo", xr, w.half().double().view(cout, cin)) + bias.double() + (res.double() if use_res else 0)
    d = (out.double() - ref).abs()
    bad = ~(d < 1e-2)
    rows = bad.any(-1).view(B, -1)
    print(f"B{B} {H}x{W} {cin}->{cout} ws={use_ws} cs={use_cs} res={use_res}: max err {float(d[~torch.isnan(d)].max()) if (~torch.isnan(d)).any() else float('nan'):.3g} nan {int(torch.isnan(out).sum())} bad rows/sample {rows.sum(1).tolist()} first bad {[int(r.nonzero()[0]) if r.any() else -1 for r in rows]}")
for args in [(3, 10, 10, 1024, 1024, True, True, True), (3, 10, 10, 1024, 1024, False, True, True), (3, 10, 10, 1024, 1024, True, False, True), (3, 10, 10, 1024, 1024, True, False, False),
             (3, 10, 10, 256, 128, True, True, True), (3, 10, 10, 256, 128, False, False, False), (2, 10, 10, 1024, 1024, True, True, True), (4, 10, 10, 1024, 1024, True, True, True),
             (3, 12, 12, 1024, 1024, True, True, True), (3, 8, 8, 1024, 1024, True, True, True), (5, 6, 6, 512, 512, True, True, True)]:
    run(*args)
