#!/usr/bin/env python3
"""Timing of the fp32 Linears / GEMMs of the embedding path (stedm_linear, stedm_gemm_f32) at a training batch of 64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stedm_amd import ops
dev = torch.device("cuda:0")
def tm(f, n=50):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for K, N, what in [(128, 512, "time_embed[0]"), (512, 512, "time_embed[2]"), (512, 10368, "emb_layers stacked"), (512, 1024, "style emb")]:
    x = torch.randn(B, K, device=dev); wt = torch.randn(K, N, device=dev); b = torch.randn(N, device=dev); out = torch.empty(B, N, device=dev)
    print(f"linear B={B} K={K} N={N} ({what}): {tm(lambda: ops.linear(x, wt, b, out, act_in=1)):.1f} us")
ws = torch.empty(1 << 24, device=dev)
for M, N, K, ta, what in [(10368, 512, B, True, "dWcat = dE^T S"), (B, 512, 10368, False, "dS = dE Wcat (split K)"), (512, 512, B, True, "dW2 = demb^T h1"),
                          (B, 512, 512, False, "dh1 = demb W2"), (512, 128, B, True, "dW0"), (1024, 512, B, True, "style dW"), (B, 512, 1024, False, "style dS")]:
    A = torch.randn((K, M) if ta else (M, K), device=dev); Bm = torch.randn(K, N, device=dev); C = torch.empty(M, N, device=dev)
    print(f"gemm_f32 M={M} N={N} K={K} ta={int(ta)} ({what}): {tm(lambda: ops.gemm_f32(A, ta, Bm, False, C, ws=ws)):.1f} us")
