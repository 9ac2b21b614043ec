"""Optimizer-step timing on the north-star U-Net (AdamW + EMA + the weight re-packs of the next forward / backward)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stedm_amd.utils import prng
from stedm_amd.unet import UNetModel
from stedm_amd.train import UNetTrainer

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


def step():
    tr._grads_ready = True
    tr.optimizer_step()
    m._prepare()                                                     # the forward's re-packs
    tr._dplan.run(versions=tuple(p._version for p in m.parameters()))   # the backward's


for _ in range(3):
    step()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    step()
e1.record(); torch.cuda.synchronize()
print(f"optimizer step + re-packs: {e0.elapsed_time(e1) / 20 * 1000:.0f} us  (fused {tr.fuse_packs})")
