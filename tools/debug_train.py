import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stedm_amd.utils import prng
from stedm_amd.unet import UNetModel
from stedm_amd.train import UNetTrainer
from tests.golden.make_golden_grads import pick_index
tag = sys.argv[1] if len(sys.argv) > 1 else "tiny"
prec = sys.argv[2] if len(sys.argv) > 2 else "parity"
cfgs = {"tiny": (dict(image_size=16, in_channels=7, model_channels=32, out_channels=4, num_res_blocks=2, attention_resolutions=[32, 16, 8], channel_mult=[1, 2, 4], num_heads=4), 2, 16, 6),
        "ns32": (dict(image_size=32, in_channels=7, model_channels=128, out_channels=4, num_res_blocks=2, attention_resolutions=[32, 16, 8], channel_mult=[1, 4, 8], num_heads=8), 2, 32, 0)}
cfg, B, hw, seed = cfgs[tag]
fx = np.load(f"tests/golden/f14_grads_{tag}.npz")
dev = torch.device("cuda:0")
m = UNetModel(precision=prec, **cfg).eval(); prng.fill_module_(m, seed=seed); m = m.to(dev)
tr = UNetTrainer(m)
x = prng.normal(seed, f"unet.{tag}.x", (B, 7, hw, hw)).to(dev)
ctx = prng.normal(seed, f"unet.{tag}.ctx", (B, cfg["model_channels"] * 4)).to(dev)
target = prng.normal(seed, f"unet.{tag}.target", (B, 4, hw, hw)).to(dev)
t = torch.from_numpy(fx["t"]).to(dev)
loss, dx, dctx = tr.loss_and_backward(x[:, :4].contiguous(), x[:, 4:].contiguous(), t, ctx, target)
print("loss", float(loss), float(fx["loss"]))
print("dctx", float((dctx.double().cpu() - torch.from_numpy(fx["dctx"]).double()).norm() / torch.from_numpy(fx["dctx"]).double().norm()))
for name, p in m.named_parameters():
    n = float(fx[f"g.{name}.norm"])
    a = p.grad.double().reshape(-1).cpu()
    en = abs(float(a.norm()) - n) / n
    rms = n / np.sqrt(a.numel())
    ep = float(np.abs(a[torch.from_numpy(pick_index(a.numel()))].numpy() - fx[f"g.{name}.pick"]).max()) / rms
    if en > 1e-3 or ep > 1e-2:
        print(f"{name:50s} norm err {en:.2e} pick err {ep:.2e}  |g| {n:.3e}")
