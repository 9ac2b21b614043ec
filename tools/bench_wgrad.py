"""Timing of the direct 3x3 weight-gradient kernel on the training step's shapes (B = 64)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stedm_amd import ops
dev = torch.device("cuda:0")
prec = ops.Precision.parse("bf16")
shapes = [(64, 8, 8, 1024, 1024), (64, 8, 8, 2048, 1024), (64, 16, 16, 512, 512), (64, 16, 16, 1536, 512), (64, 32, 32, 128, 128), (64, 32, 32, 640, 128)]
for B, H, W, ci, co in shapes:
    x = torch.randn(B, H, W, ci, device=dev).bfloat16().view(torch.int16)
    dy = torch.randn(B, H, W, co, device=dev).bfloat16().view(torch.int16)
    ks = ops.wgrad3x3_plan(B, H, W, ci, co)
    part = torch.empty((ks * 9 * ci * co,), dtype=torch.float32, device=dev)
    for _ in range(3): ops.wgrad3x3_oihw(x, dy, part, prec)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ops.wgrad3x3_oihw(x, dy, part, prec)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    fl = 2.0 * B * H * W * 9 * ci * co
    print(f"B={B} {H}x{W} {ci}->{co} ksplit {ks}: {dt * 1e6:7.1f} us  {fl / dt / 1e12:7.1f} TFLOP/s")
