#!/usr/bin/env python3
"""MFMA rate of the LSA flash attention kernel alone (vit_set.py:52-66; T = 4098 tokens = 4096 patches of a 512^2 style image + cls + time
token, 12 heads of 64): 4 * T^2 * 64 FLOP per head. Random operands.

    python tools/bench_lsa.py [bf16|f16|parity] [B] [p_drop]
"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stedm_amd import ops


def main():
    prec = ops.Precision.parse(sys.argv[1] if len(sys.argv) > 1 else "bf16")
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    p = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    T, heads = int(os.environ.get("LSA_T", "4098")), 12      # LSA_T: another token count (4096: no 2-query tail tile)
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    Tp = (T + 127) // 128 * 128
    qkv = torch.randn(B, T, 3 * heads * 64, device=dev)
    i16 = torch.int16
    lo = prec.npass == 3
    mk = lambda shp: (torch.zeros(shp, dtype=i16, device=dev), torch.zeros(shp, dtype=i16, device=dev) if lo else None)
    q, k, v = mk((B * heads, Tp, 64)), mk((B * heads, Tp, 64)), mk((B * heads, 64, Tp))
    ops.qkv_pack(qkv, 0.125 * math.log2(math.e), q, k, v, B, T, Tp, heads, prec)
    out = mk((B, T, heads * 64))
    if prec.attn_fp8:       # MX-fp8 operands (python tools/bench_lsa.py fp8 B): the pack pass is timed separately
        u8 = torch.uint8
        q8 = torch.zeros((B * heads, Tp, 64), dtype=u8, device=dev); k8 = torch.zeros_like(q8); v8 = torch.zeros((B * heads, 64, Tp), dtype=u8, device=dev)
        qs = torch.zeros((B * heads, Tp, 2), dtype=u8, device=dev); ks = torch.zeros_like(qs); vs = torch.zeros((B * heads, Tp // 32, 64), dtype=u8, device=dev)
        q16 = qkv.to(torch.bfloat16).view(torch.int16)
        for name, fn in (("qkv_pack_mx8 (from 16-bit qkv)", lambda: ops.qkv_pack_mx8(q16, 0.125 * math.log2(math.e), q8, qs, k8, ks, v8, vs, B, T, Tp, heads, prec)),
                         ("qkv_pack (16-bit -> 16-bit, the bf16 path's)", lambda: ops.qkv_pack(q16, 0.125 * math.log2(math.e), q, k, v, B, T, Tp, heads, prec)),
                         ("lsa_flash_mx8", lambda: ops.lsa_flash_mx8(q8, qs, k8, ks, v8, vs, out[0], B, T, Tp, heads, prec))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 10 * 1e3
            fl = 4.0 * T * T * 64 * heads * B
            print(f"{name} B={B} T={T}: {us:.1f} us per call" + (f", {fl / us / 1e6:.1f} TFLOP/s ({fl / us / 1e6 / 2500 * 100:.1f} % of the 2.5 PF bf16 dense peak)" if "flash" in name else ""), flush=True)
        return
    run = (lambda: ops.lsa_flash_drop(q, k, v, out, B, T, Tp, heads, prec, p, 1234, 1)) if p > 0 else (lambda: ops.lsa_flash(q, k, v, out, B, T, Tp, heads, prec))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    n = 10
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    fl = 4.0 * T * T * 64 * heads * B
    print(f"lsa_flash {prec.label} B={B} T={T} p={p}: {us:.1f} us per call, {fl / us / 1e6:.1f} TFLOP/s ({fl / us / 1e6 / 2500 * 100:.1f} % of the 2.5 PF dense peak)"
          "", flush=True)


if __name__ == "__main__":
    main()
