#!/usr/bin/env python3
"""BASELINE config 1 on the HIP path: NS32, batch B (default 1), DDIM-20 + CFG 1.5, the whole loop through `sample_log` (hipGraph replay).
    python tools/bench_small.py [B] [precision]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    ld = bench.build_model(dev, prec)
    g = torch.Generator().manual_seed(3)
    xT = torch.randn(B, 4, 32, 32, generator=g).to(dev)
    cc = (torch.randn(B, 3, 32, 32, generator=g) > 0).float().to(dev)
    ctx = torch.randn(B, 512, generator=g).to(dev); ctx_u = torch.randn(1, 512, generator=g).repeat(B, 1).to(dev)
    cond = {"c_concat": [cc], "c_crossattn": [ctx]}; unc = {"c_concat": [cc], "c_crossattn": [ctx_u]}
    run = lambda: ld.sample_log(cond, B, True, 20, eta=0.0, x_T=xT, unconditional_conditioning=unc, unconditional_guidance_scale=1.5, log_every_t=1000)[0]
    run(); run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        s = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"DDIM-20 + CFG loop, batch {B}, {prec}: {dt * 1e3:.2f} ms per loop = {dt * 50:.3f} ms per step; finite {bool(torch.isfinite(s).all())}", flush=True)


if __name__ == "__main__":
    main()
