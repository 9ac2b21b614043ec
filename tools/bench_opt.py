"""Optimizer kernels on the north-star U-Net: the fused AdamW + EMA + weight-pack pass against the plain AdamW + EMA kernel and the two
stedm_pack_frag_multi launches it replaces (HIP events over 20 back-to-back launches each)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stedm_amd.utils import prng
from stedm_amd.unet import UNetModel
from stedm_amd.train import UNetTrainer
from stedm_amd import ops
NS32 = dict(image_size=32, in_channels=7, model_channels=128, out_channels=4, num_res_blocks=2, attention_resolutions=[32, 16, 8], channel_mult=[1, 4, 8], num_heads=8)
dev = torch.device("cuda:0")
m = UNetModel(precision="bf16", **NS32).eval(); prng.fill_module_(m, seed=0); m = m.to(dev)
tr = UNetTrainer(m, lr=1e-5)
B = 8
g = torch.Generator(device="cpu").manual_seed(1)
x = torch.randn(B, 4, 32, 32, generator=g).to(dev); cc = torch.randn(B, 3, 32, 32, generator=g).to(dev)
ctx = torch.randn(B, 512, generator=g).to(dev); tgt = torch.randn(B, 4, 32, 32, generator=g).to(dev)
t = torch.randint(0, 1000, (B,), generator=g).to(dev)
for _ in range(2):
    tr.train_step(x, cc, t, ctx, tgt)
st = tr._opt
fu = tr._fused_opt()
def k_fused():
    ops.adamw_ema_pack(fu["descs"], fu["n"], fu["blocks"], 1e-5, 0.9, 0.999, 1e-8, 0.01, 3, 0.9999, 1.0)
def k_plain():
    ops.adamw_ema(st["table"], st["ct"], st["co"], 1e-5, 0.9, 0.999, 1e-8, 0.01, 3, 0.9999)
def k_pack():
    m._plan.run(); tr._dplan.run()
for name, f in (("fused adamw+pack", k_fused), ("plain adamw (all tensors)", k_plain), ("pack_frag_multi x2", k_pack)):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 20 * 1000:.0f} us")
n = sum(p.numel() for p in st["params"])
print("parameters", n)
