"""GPU tier: the frozen VQ-f4 first stage — stedm_amd.vq.VQModelInterface (reference module surface:
ldm.models.autoencoder.VQModelInterface over model.Encoder / model.Decoder) against the reference's own Encoder / Decoder outputs
(tests/golden/f15_vq_*.npz) and the CPU oracle; the nearest-codebook indices (integer work) bit-exact against the oracle's pinned-down
arithmetic. Tolerance 1e-3 relative in parity mode; the single-product modes are reported."""
import numpy as np
import pytest
import torch

from stedm_amd.utils import prng
from tests.golden.summary import check_summary

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

DD_TINY = dict(double_z=False, z_channels=3, resolution=64, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 4], num_res_blocks=1,
               attn_resolutions=[], dropout=0.0)
DD_F4 = dict(double_z=False, z_channels=3, resolution=512, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4], num_res_blocks=2,
             attn_resolutions=[], dropout=0.0)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def build(dd, dev, precision="parity", embed_dim=3, n_embed=8192, seed=15):
    from stedm_amd.vq import VQModelInterface
    m = VQModelInterface(embed_dim=embed_dim, n_embed=n_embed, ddconfig=dd, lossconfig={"target": "torch.nn.Identity"}, precision=precision).eval()
    prng.fill_module_(m, seed=seed)
    with torch.no_grad():
        m.quantize.embedding.weight.copy_(prng.normal(seed, "quantize.embedding.weight", (n_embed, embed_dim)) * 0.7)
    return m.to(dev)


def rel(a, b):
    a = a.double().cpu(); b = torch.as_tensor(np.asarray(b)).double()
    return float((a - b).abs().max()) / (float(b.std()) + 1e-12)


def test_state_dict_names_match_reference_layout():
    """same names / shapes as ldm.models.autoencoder.VQModelInterface's state dict (vq-f4.ckpt loads with load_state_dict)"""
    from oracle import vq as ovq
    from stedm_amd.vq import VQModelInterface
    m = VQModelInterface(embed_dim=3, n_embed=8192, ddconfig=DD_F4, lossconfig={"target": "torch.nn.Identity"})
    sh = ovq.shapes(ovq.VQConfig())
    sd = m.state_dict()
    assert set(sd) == set(sh)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(sh[k]), k
    assert not any(p.requires_grad for p in m.parameters())


@pytest.mark.parametrize("e_dim,n_e,B,H", [(3, 8192, 2, 32), (4, 8192, 3, 16), (3, 100, 1, 8)])
def test_nearest_codebook_indices_bit_exact(dev, e_dim, n_e, B, H):
    """integer result: indices equal the oracle's (pinned-down fp32 arithmetic, first index on ties) bit for bit, incl. duplicated
    codebook entries (forced ties) and latents that sit exactly on an entry; the straight-through value z + (e - z) bitwise."""
    from oracle import vq as ovq
    from stedm_amd import ops
    cb = prng.normal(17, f"cb.{e_dim}.{n_e}", (n_e, e_dim)) * 0.6
    cb[n_e // 2: n_e // 2 + 5] = cb[3:8]                                   # duplicates: index 3..7 must win
    z = prng.normal(17, f"z.{e_dim}.{B}.{H}", (B, e_dim, H, H))
    z[0, :, 0, :5] = cb[10:15].t()                                         # exactly on entries
    idx_ref, zq_ref = ovq.quantize(cb, z)
    idx, zq = ops.vq_nearest(z.to(dev), cb.to(dev).contiguous())
    assert idx.dtype == torch.int64 and torch.equal(idx.cpu(), idx_ref)
    assert torch.equal(zq.cpu(), zq_ref)
    assert idx_ref[0, 0, :5].tolist() == [10, 11, 12, 13, 14]


@pytest.mark.parametrize("tag,dd,B,side", [("tiny", DD_TINY, 2, 64), ("f4", dict(DD_F4, resolution=128), 1, 128)])
def test_encoder_decoder_vs_reference_golden(dev, golden, tag, dd, B, side):
    fx = golden(f"f15_vq_{tag}")
    m = build(dd, dev)
    m._prepare()
    x = prng.uniform(15, f"vq.{tag}.x", (B, 3, side, side)).to(dev)
    z = prng.normal(15, f"vq.{tag}.z", (B, 3, side // 4, side // 4)).to(dev)
    m._cs = {}
    ze = m._encoder(x)
    e1 = rel(ze, fx["enc_out"])
    m._cs = {}
    yd = m._decoder(z)
    print(f"[VQ {tag} parity] encoder rel err vs reference golden {e1:.3e}")
    assert e1 < 1e-3
    e2 = check_summary(yd, fx, "dec_out", 1e-3, tag)
    print(f"[VQ {tag} parity] decoder sampled rel err vs reference golden {e2:.3e}")
    if "dec_out" in fx.files:
        assert rel(yd, fx["dec_out"]) < 1e-3
    for precision in ("f16", "bf16"):
        m.set_precision(precision)
        m._prepare(); m._cs = {}
        a = rel(m._encoder(x), fx["enc_out"])
        m._cs = {}
        yd2 = m._decoder(z)
        b = float((yd2.double() - yd.double()).norm() / yd.double().norm())
        print(f"[VQ {tag} {precision}] encoder max/std vs golden {a:.3e}; decoder rel-L2 vs parity mode {b:.3e}")
        assert a < 0.2 and b < 0.1
    m.set_precision("parity")


def test_interface_encode_decode_vs_oracle(dev):
    """VQModelInterface.encode (encoder + quant_conv, no quantisation) and .decode (quantise + post_quant_conv + decoder,
    autoencoder.py:269-282) against the oracle; 4-channel latents (the synthetic NS32 shape family) as well as the shipped 3."""
    from oracle import vq as ovq
    for zc in (3, 4):
        dd = dict(DD_TINY, z_channels=zc)
        cfg = ovq.VQConfig(ch=32, num_res_blocks=1, z_channels=zc, embed_dim=zc, n_embed=512)
        m = build(dd, dev, embed_dim=zc, n_embed=512, seed=18)
        P = prng.fill_state_dict(ovq.shapes(cfg), 18)
        P["quantize.embedding.weight"] = prng.normal(18, "quantize.embedding.weight", (512, zc)) * 0.7
        x = prng.uniform(18, "if.x", (2, 3, 64, 64))
        h = prng.normal(18, f"if.h{zc}", (2, zc, 16, 16)) * 0.8
        assert rel(m.encode(x.to(dev)), ovq.vq_encode(P, cfg, x)) < 1e-3
        assert rel(m.decode(h.to(dev)), ovq.vq_decode(P, cfg, h)) < 1e-3
        assert rel(m.decode(h.to(dev), force_not_quantize=True), ovq.vq_decode(P, cfg, h, force_not_quantize=True)) < 1e-3
        q, _, (_, _, ind) = m.quantize(h.to(dev))
        assert torch.equal(ind.cpu(), ovq.quantize(P["quantize.embedding.weight"], h)[0].reshape(-1))


def test_predict_step_with_first_stage_end_to_end(dev):
    """LDM_Diffusion.predict_step (modules/ldm_diffusion.py:76-107) through stedm_amd.ldm_module with the first stage given as the
    reference's YAML dict ({target: ldm.models.autoencoder.VQModelInterface, params}): sampled latents -> decode_first_stage -> uint8
    images; decoded floats against the oracle's decoder on the same latents, uint8 by the reference's truncating cast."""
    from oracle import vq as ovq
    from stedm_amd.latent_diffusion import predict_latents
    from stedm_amd.ldm_module import LDM_Diffusion
    from tests.test_gpu_train import MODULE_UNET, _module_batches, _module_cfg
    cfg = _module_cfg()
    cfg["diffusion"]["first_stage_config"] = {"target": "ldm.models.autoencoder.VQModelInterface",
                                              "params": dict(embed_dim=4, n_embed=256, ddconfig=dict(DD_TINY, z_channels=4), lossconfig={"target": "torch.nn.Identity"})}
    mod = LDM_Diffusion(cfg)
    zm = mod._model
    assert type(zm.first_stage_model).__name__ == "VQModelInterface"
    prng.fill_module_(zm.model.diffusion_model, seed=6); prng.fill_module_(zm.cond_stage_model, seed=9); prng.fill_module_(zm.agg_block, seed=51)
    prng.fill_module_(zm.first_stage_model, seed=19)
    mod = mod.to(dev).eval()
    batch = _module_batches(dev, 1)[0]
    torch.manual_seed(3)
    img, seg = mod.predict_step(batch, 0)
    assert img.shape == (2, 64, 64, 3) and img.dtype == np.uint8 and seg.shape == (2, 64, 64) and seg.max() <= 1
    # same latents again (deterministic sampler given x_T) -> oracle decode
    torch.manual_seed(3)
    lb = mod.prepare_batch(batch)
    lat = predict_latents(zm, lb, ddim_steps=4, eta=0.0, cfg_scale=1.5, style_sampling="mp")
    ocfg = ovq.VQConfig(ch=32, num_res_blocks=1, z_channels=4, embed_dim=4, n_embed=256)
    P = {k: v.detach().cpu() for k, v in zm.first_stage_model.state_dict().items()}
    ref = ovq.vq_decode(P, ocfg, lat.cpu())
    dec = zm.decode_first_stage(lat)
    assert rel(dec, ref) < 1e-3
    ref8 = ((np.clip(ref.permute(0, 2, 3, 1).numpy(), -1, 1) + 1) * 127.5).astype(np.uint8)
    diff = np.abs(img.astype(np.int32) - ref8.astype(np.int32))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.02          # truncating cast: a float 1e-5 apart may land on the other side of an integer
    assert np.array_equal(seg, torch.argmax(lb["segmentation"], dim=-1).cpu().numpy().astype(np.uint8))


def test_decoder_at_256_pixel_rows_vs_oracle(dev):
    """the shipped vq-f4 architecture decoding a 64^2 latent to a 256^2 image: its last level runs on 256-pixel rows (one tile per row),
    the level before on 128 — against the oracle's decoder (summary of the output)."""
    from oracle import vq as ovq
    cfg = ovq.VQConfig()
    m = build(dict(DD_F4, resolution=256), dev)
    P = prng.fill_state_dict(ovq.shapes(cfg), 15)
    z = prng.normal(15, "vq.w256.z", (1, 3, 64, 64))
    m._prepare(); m._cs = {}
    y = m._decoder(z.to(dev))
    ref = ovq.decoder(P, cfg, z)
    assert tuple(y.shape) == (1, 3, 256, 256)
    err = rel(y, ref)
    print(f"[VQ f4 decoder 64^2 -> 256^2, parity] rel err vs oracle {err:.3e}")
    assert err < 1e-3
    m.set_precision("bf16"); m._prepare(); m._cs = {}
    y2 = m._decoder(z.to(dev))
    l2 = float((y2.double().cpu() - ref.double()).norm() / ref.double().norm())
    print(f"[VQ f4 decoder 64^2 -> 256^2, bf16] rel-L2 vs oracle {l2:.3e}")
    assert l2 < 0.1


def test_decoder_reference_native_512_vs_oracle(dev):
    """the reference's own size: vq-f4.yaml decoding a 128^2 x 3 latent to a 512^2 image (conf/diffusion/ldm_based.yaml image_size 128,
    conf/data patch_size 512): 16 384-token middle attention, 512-pixel rows at the last level (two tiles per row), the 256 -> 512 upsample
    through the materialised plane. One CPU oracle decode (~1.3 TFLOP); parity mode < 1e-3, bf16 reported."""
    from oracle import vq as ovq
    cfg = ovq.VQConfig()
    m = build(dict(DD_F4, resolution=512), dev)
    P = prng.fill_state_dict(ovq.shapes(cfg), 15)
    z = prng.normal(15, "vq.w512.z", (1, 3, 128, 128))
    ref = ovq.decoder(P, cfg, z)
    m._prepare(); m._cs = {}
    y = m._decoder(z.to(dev))
    assert tuple(y.shape) == (1, 3, 512, 512)
    err = rel(y, ref)
    m.set_precision("bf16"); m._prepare(); m._cs = {}
    y2 = m._decoder(z.to(dev))
    l2 = float((y2.double().cpu() - ref.double()).norm() / ref.double().norm())
    print(f"[VQ f4 decoder 128^2 -> 512^2] parity rel err vs oracle {err:.3e}; bf16 rel-L2 {l2:.3e}")
    assert err < 1e-3 and l2 < 0.1
