"""Where the torch-side glue launches of a training step come from: runs one step under torch.profiler (with_stack) and lists, for the
ATen ops that launch fills / copies / elementwise kernels, the stedm_amd call sites by count.    python tools/trace_glue.py"""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stedm_amd.utils import prng
from stedm_amd.unet import UNetModel
from stedm_amd.train import UNetTrainer

NS32 = dict(image_size=32, in_channels=7, model_channels=128, out_channels=4, num_res_blocks=2, attention_resolutions=[32, 16, 8], channel_mult=[1, 4, 8], num_heads=8)
dev = torch.device("cuda:0")
m = UNetModel(precision="bf16", **NS32).eval(); prng.fill_module_(m, seed=0); m = m.to(dev)
tr = UNetTrainer(m, lr=1e-5)
B = 64
g = torch.Generator(device="cpu").manual_seed(1)
x = torch.randn(B, 4, 32, 32, generator=g).to(dev); cc = torch.randn(B, 3, 32, 32, generator=g).to(dev)
ctx = torch.randn(B, 512, generator=g).to(dev); tgt = torch.randn(B, 4, 32, 32, generator=g).to(dev)
t = torch.randint(0, 1000, (B,), generator=g).to(dev)
for _ in range(3):
    tr.train_step(x, cc, t, ctx, tgt)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_step(x, cc, t, ctx, tgt)
    torch.cuda.synchronize()
sites = collections.Counter()
dur = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue                       # outermost ATen op only
    dt = ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total
    if dt <= 0:
        continue                       # launched no kernel
    site = next((s for s in ev.stack if "stedm_amd" in s), ev.stack[0] if ev.stack else "?")
    sites[(ev.name, site.strip())] += 1
    dur[(ev.name, site.strip())] += dt
tot = sum(dur.values())
print(f"torch-side kernels of one training step: {sum(sites.values())} launches, {tot:.0f} us of device time")
for k, n in sorted(sites.items(), key=lambda kv: -dur[kv[0]])[:60]:
    print(f"{dur[k]:8.0f} us {n:4d} x {k[0]:28s} {k[1][-110:]}")
