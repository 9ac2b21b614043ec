"""GPU tier (-m gpu): style path — set-ViT encoder (patch embed, LSA flash attention with diagonal mask, MLP, pooled
head), aggregation blocks and the layout SpatialRescaler — against the reference's golden vectors and the CPU oracle."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from stedm_amd.utils import prng

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def rel(a, b):
    a = a.double().cpu(); b = torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max()) / (float(b.std()) + 1e-12)


def make_svit(dev, img, ns, precision="parity"):
    from stedm_amd.style import sViT
    m = sViT(image_size=img, patch_size=8, num_classes=512, dim=256, depth=6, heads=12, mlp_dim=256, pool="mean", channels=3,
             dropout=0.1, emb_dropout=0.1, ns=ns, t_dim=256, precision=precision).eval()
    prng.fill_module_(m, seed=7)
    for l, (attn, _ff) in enumerate(m.transformer.layers):
        attn.fn.temperature.fill_(float(np.log(64 ** -0.5)) + 0.05 * l)
    return m.to(dev)


def test_svit_state_dict_names(dev):
    from oracle import style as ost
    from stedm_amd.style import sViT
    m = sViT(image_size=64, patch_size=8, num_classes=512, dim=256, depth=6, heads=12, mlp_dim=256, pool="mean", ns=4)
    shapes = ost.svit_shapes(ost.SViTConfig(image_size=64, ns=4))
    sd = m.state_dict()
    assert set(sd) == set(shapes)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k


@pytest.mark.parametrize("tag,img,ns,B", [("i64_ns1", 64, 1, 2), ("i64_ns4", 64, 4, 2)])
def test_svit_vs_reference_golden(dev, golden, tag, img, ns, B):
    fx = golden("f7_svit")
    m = make_svit(dev, img, ns)
    x = prng.uniform(7, f"svit.{tag}.img", (B, ns, img, img, 3)).to(dev)
    y = m(x)
    err = rel(y, fx[tag])
    print(f"[sViT {tag}] parity-mode rel err vs reference golden: {err:.3e}")
    assert err < 1e-3


def test_svit_512_vs_reference_golden(dev, golden):
    """Full-size style encoder: 4 images of 512x512 -> 4098 tokens (the reference materialises 12 x 4098^2 logits)."""
    fx = golden("f7_svit")
    m = make_svit(dev, 512, 4)
    x = prng.uniform(7, "svit.i512_ns4.img", (1, 4, 512, 512, 3)).to(dev)
    y = m(x)
    err = rel(y, fx["i512_ns4"])
    print(f"[sViT 512 ns4] parity-mode rel err vs reference golden: {err:.3e}")
    assert err < 1e-3


@pytest.mark.parametrize("precision,tol", [("f16", 1e-2), ("bf16", 8e-2)])
def test_svit_fast_modes_reported(dev, golden, precision, tol):
    fx = golden("f7_svit")
    m = make_svit(dev, 64, 4, precision)
    x = prng.uniform(7, "svit.i64_ns4.img", (2, 4, 64, 64, 3)).to(dev)
    err = rel(m(x), fx["i64_ns4"])
    print(f"[sViT i64_ns4 {precision}] rel err vs reference golden: {err:.3e}")
    assert err < tol


@pytest.mark.parametrize("B,T,heads", [(2, 66, 12), (1, 130, 2), (1, 300, 3)])
def test_lsa_flash_vs_oracle(dev, B, T, heads):
    """LSA core (vit_set.py:56-66) alone: logits * exp(tau), diagonal masked, softmax, @ v."""
    from stedm_amd import ops
    prec = ops.Precision.parse("parity")
    qkv = prng.normal(40, "lsa.qkv", (B, T, 3 * heads * 64))
    tau = math.exp(math.log(64 ** -0.5) + 0.1)
    q, k, v = (t.reshape(B, T, heads, 64).permute(0, 2, 1, 3) for t in qkv.chunk(3, dim=-1))
    dots = torch.matmul(q, k.transpose(-1, -2)) * tau
    dots = dots.masked_fill(torch.eye(T, dtype=torch.bool), -torch.finfo(dots.dtype).max)
    ref = torch.matmul(dots.softmax(dim=-1), v).permute(0, 2, 1, 3).reshape(B, T, heads * 64)
    Tp = ((T + 127) // 128) * 128
    i16 = torch.int16
    mk = lambda shp: (torch.zeros(shp, dtype=i16, device=dev), torch.zeros(shp, dtype=i16, device=dev))
    qd, kd, vd = mk((B * heads, Tp, 64)), mk((B * heads, Tp, 64)), mk((B * heads, 64, Tp))
    ops.qkv_pack(qkv.to(dev).contiguous(), tau * math.log2(math.e), qd, kd, vd, B, T, Tp, heads, prec)   # lsa_flash works in log2
    od = mk((B, T, heads * 64))
    ops.lsa_flash(qd, kd, vd, od, B, T, Tp, heads, prec)
    got = od[0].view(torch.float16).float() + od[1].view(torch.float16).float()
    assert rel(got, ref) < 2e-4


def test_agg_blocks_vs_reference_golden(dev, golden):
    import types
    from stedm_amd import style as st
    fx = golden("f8_agg")

    class StandIn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.proj = torch.nn.Linear(3 * 4 * 4, 512)

        def forward(self, x):
            return self.proj(F.adaptive_avg_pool2d(x, 4).flatten(1))

    samp = types.SimpleNamespace(name="mp", num_patches=4)
    sty = prng.uniform(8, "agg.style", (2, 4, 16, 16, 3)).to(dev)
    for name, cls in (("mean", st.Agg_Mean), ("max", st.Agg_Max), ("linear", st.Agg_Linear)):
        m = cls(samp, StandIn()).eval()
        sd = {k: prng.fill_value(8, k, v.shape) for k, v in m.state_dict().items() if not k.startswith("_")}
        m.load_state_dict(sd, strict=False)
        y = m.to(dev)(sty)
        assert rel(y, fx[name]) < 2e-5, name
    assert torch.equal(st.Agg_None(samp, None)(sty).cpu(), torch.from_numpy(fx["none"]))


def test_spatial_rescaler_vs_reference_golden(dev, golden):
    from stedm_amd.style import SpatialRescaler
    fx = golden("f9_rescaler")
    m = SpatialRescaler(n_stages=2, in_channels=2, out_channels=3).eval()
    prng.fill_module_(m, seed=9)
    seg = (prng.uniform(9, "resc.seg", (2, 2, 64, 64)) > 0).float()
    y = m.to(dev)(seg.to(dev))
    assert rel(y, fx["y"]) < 2e-6
    assert rel(m.encode(seg.to(dev)), fx["y"]) < 2e-6
