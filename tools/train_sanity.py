"""Sanity run of the training step: 40 optimizer steps at batch 64 on fresh synthetic batches (fixed seed); the loss must stay finite
and fall from its initial value, parameters must stay finite."""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from stedm_amd.utils import prng
from stedm_amd.unet import UNetModel
from stedm_amd.train import UNetTrainer
NS32 = dict(image_size=32, in_channels=7, model_channels=128, out_channels=4, num_res_blocks=2, attention_resolutions=[32, 16, 8], channel_mult=[1, 4, 8], num_heads=8)
dev = torch.device("cuda:0")
m = UNetModel(precision="bf16", **NS32).eval(); prng.fill_module_(m, seed=0); m = m.to(dev)
tr = UNetTrainer(m, lr=1e-4, weight_decay=0.01)
g = torch.Generator(device="cpu").manual_seed(1)
B = 64
losses = []
for it in range(40):
    x = torch.randn(B, 4, 32, 32, generator=g).to(dev); cc = (torch.randn(B, 3, 32, 32, generator=g) > 0).float().to(dev)
    ctx = torch.randn(B, 512, generator=g).to(dev); noise = torch.randn(B, 4, 32, 32, generator=g).to(dev)
    t = torch.randint(0, 1000, (B,), generator=g).to(dev)
    # the input is the noise itself and the target is that noise (the t -> T limit of q_sample)
    loss = tr.train_step(noise, cc, t, ctx, noise)
    losses.append(float(loss))
print("losses:", " ".join(f"{v:.4f}" for v in losses[::3]))
assert all(v == v and v < 10 for v in losses), "non-finite or exploding loss"
assert losses[-1] < losses[0] - 0.05, "the loss did not fall"
pmax = max(float(p.detach().abs().max()) for p in m.parameters())
print("max |param| after 40 steps:", pmax, "finite:", all(bool(torch.isfinite(p).all()) for p in m.parameters()))
