#!/usr/bin/env python3
"""decode_first_stage alone (VQ-f4 architecture: quantise, post_quant_conv, Decoder 32^2 -> 128^2) at the bench's batch.
    python tools/bench_vq.py [bf16|f16|parity] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stedm_amd.utils import prng
from stedm_amd.vq import VQModelInterface

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
m = VQModelInterface(embed_dim=4, n_embed=8192, lossconfig={"target": "torch.nn.Identity"}, precision=prec,
                     ddconfig=dict(double_z=False, z_channels=4, resolution=128, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4], num_res_blocks=2,
                                   attn_resolutions=[], dropout=0.0)).eval()
prng.fill_module_(m, seed=53)
m = m.to(dev)
z = torch.randn(B, 4, 32, 32, device=dev)
for _ in range(2):
    y = m.decode(z)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
n = 5
e0.record()
for _ in range(n):
    y = m.decode(z)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(f"VQ-f4 decode {prec} B={B}: {ms:.2f} ms ({B / ms * 1e3:.0f} images/s), out {tuple(y.shape)}, finite {bool(torch.isfinite(y).all())}")
