"""GroupNorm-apply launches of one CFG denoising step (B = 64) in execution order: shapes and bytes (for reading a kernel trace)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stedm_amd import ops

dev = torch.device("cuda:0")
ld = bench.build_model(dev, "bf16", use_graph=False)
xT, cond, unc = bench.synth_inputs(dev, 64, 0)
rec = []
orig = ops.gn_apply16c
def wrapped(x1, cs1, x2, cs2, out_hi, *a, **k):
    B = x1.shape[0]; C = x1.shape[-1] + (0 if x2 is None else x2.shape[-1]); HW = x1.numel() // (B * x1.shape[-1])
    rec.append((B, HW, x1.shape[-1], 0 if x2 is None else x2.shape[-1]))
    return orig(x1, cs1, x2, cs2, out_hi, *a, **k)
ops.gn_apply16c = wrapped
import stedm_amd.unet as U
with torch.no_grad():
    ld.sample_log(cond, 64, True, 1, eta=0.0, unconditional_conditioning=unc, unconditional_guidance_scale=1.5)
print(json.dumps(rec))
