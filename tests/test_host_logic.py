"""CPU tier: product host logic (schedules, DDIM tables, module surface) against the reference's golden vectors."""
import numpy as np
import torch

from stedm_amd import schedule as sch


def test_schedule_matches_reference_bits(golden):
    fx = golden("f2_schedule")
    ns = sch.NoiseSchedule.make(1000, 0.0015, 0.0205)
    assert np.array_equal(ns.betas, fx["betas_f32"])
    assert np.array_equal(ns.alphas_cumprod, fx["alphas_cumprod_f32"])
    assert ns.alphas_cumprod_prev[0] == np.float32(1.0)
    for S, n in ((20, 20), (50, 50), (128, 143)):
        ts = sch.make_ddim_timesteps(S)
        assert ts.dtype == np.int64 and ts.shape[0] == n
        assert np.array_equal(ts, fx[f"ts_{S}"])            # integer indexing: bit-exact
        for eta in (0.0, 1.0):
            tb = sch.make_ddim_tables(ns.alphas_cumprod, S, eta)
            f32 = lambda a: np.asarray(a, dtype=np.float64).astype(np.float32)
            assert np.array_equal(tb.alphas, f32(fx[f"a_{S}_{eta}"]))
            assert np.array_equal(tb.alphas_prev, f32(fx[f"ap_{S}_{eta}"]))
            assert np.array_equal(tb.sigmas, f32(fx[f"sig_{S}_{eta}"]))
            assert np.array_equal(tb.sqrt_one_minus_alphas, np.sqrt(np.float32(1.0) - tb.alphas))
            assert tb.coef_table().shape == (n, 4) and tb.coef_table().dtype == np.float32


def test_unet_module_surface_cpu():
    """Construction, state-dict names and zero-initialised layers follow the reference (no GPU needed)."""
    from oracle import unet as ou
    from stedm_amd.unet import UNetModel
    m = UNetModel(image_size=16, in_channels=7, model_channels=32, out_channels=4, num_res_blocks=2,
                  attention_resolutions=[32, 16, 8], channel_mult=[1, 2, 4], num_heads=4)
    plan = ou.build_plan(ou.UNetConfig(image_size=16, model_channels=32, channel_mult=(1, 2, 4), num_heads=4))
    sd = m.state_dict()
    assert set(sd) == set(plan.shapes)
    for k in ("input_blocks.1.0.out_layers.3.weight", "middle_block.2.proj_out.weight", "out.2.weight"):
        assert float(sd[k].abs().max()) == 0.0            # zero_module (openaimodel.py:242-244,334,732)
    # reference raises TypeError when ds hits attention_resolutions (openaimodel.py:580-590)
    import pytest
    with pytest.raises(TypeError):
        UNetModel(image_size=16, in_channels=7, model_channels=32, out_channels=4, num_res_blocks=2,
                  attention_resolutions=[1], channel_mult=[1, 2], num_heads=4)
    # fails loudly on CPU: no fallback
    from stedm_amd._lib import StedmHipError
    with pytest.raises(StedmHipError):
        m(torch.zeros(1, 7, 16, 16), torch.zeros(1, dtype=torch.long), context=torch.zeros(1, 128))


def test_reference_checkpoint_key_layout_roundtrip():
    """A Lightning checkpoint of the reference (`_model.` prefix, LitEma buffers named without dots, ema.py:17-21) loads into the
    module surface; with use_ema the shadows replace the U-Net weights (ema_scope, ddpm.py:174-188)."""
    from stedm_amd.latent_diffusion import LatentDiffusion
    from stedm_amd.style import SpatialRescaler
    from stedm_amd.unet import UNetModel
    from stedm_amd.utils import prng
    tiny = dict(image_size=16, in_channels=7, model_channels=32, out_channels=4, num_res_blocks=2,
                attention_resolutions=[32, 16, 8], channel_mult=[1, 2, 4], num_heads=4)

    def make(seed):
        unet = UNetModel(**tiny)
        prng.fill_module_(unet, seed=seed)
        return LatentDiffusion(unet, linear_start=0.0015, linear_end=0.0205, loss_type="l1", image_size=16, channels=4,
                               conditioning_key="hybrid", cond_stage_config=SpatialRescaler(n_stages=2, in_channels=2, out_channels=3))

    src, dst = make(1), make(2)
    sd = {"_model." + k: v.clone() for k, v in src.state_dict().items()}
    ema_vals = {}
    for name, p in src.model.named_parameters():            # what LitEma(self.model) would hold
        ema_vals[name] = p.detach() * 0.5
        sd["_model.model_ema." + name.replace(".", "")] = ema_vals[name]
    sd["_model.model_ema.decay"] = torch.tensor(0.9999)
    sd["_model.model_ema.num_updates"] = torch.tensor(7, dtype=torch.int)
    sd["_model.first_stage_model.encoder.conv_in.weight"] = torch.zeros(3)     # first stage not supplied: skipped, not an error
    missing, unexpected = dst.load_reference_state_dict({"state_dict": sd})
    assert not missing and not unexpected
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k
    assert dst._ema_num_updates == 7 and set(dst._ema_loaded) == set(ema_vals)
    dst.load_reference_state_dict({"state_dict": sd}, use_ema=True)
    for name, p in dst.model.named_parameters():
        assert torch.equal(p.detach(), ema_vals[name]), name
    out = dst.reference_state_dict()
    assert all(k.startswith("_model.") for k in out) and "_model.model.diffusion_model.out.2.weight" in out


def test_training_seam_members_have_the_reference_signatures():
    """What modules/ldm_diffusion.py:63-73, 110-115 calls on the model (SURVEY §8b seam 2): names and positional arguments of
    ddpm.py:345-371, 479-494, 868-882."""
    import inspect
    from stedm_amd.latent_diffusion import LatentDiffusion, S_ZSS_DM
    sig = lambda f: list(inspect.signature(f).parameters)
    assert sig(LatentDiffusion.training_step) == ["self", "batch", "batch_idx"]
    assert sig(LatentDiffusion.shared_step)[:2] == ["self", "batch"]
    assert sig(LatentDiffusion.forward)[:3] == ["self", "x", "c"]
    assert sig(LatentDiffusion.on_train_batch_start)[:4] == ["self", "batch", "batch_idx", "dataloader_idx"]
    assert sig(LatentDiffusion.p_losses)[:5] == ["self", "x_start", "cond", "t", "noise"]
    assert sig(S_ZSS_DM.get_input)[:5] == ["self", "batch", "k", "cond_key", "bs"]
    for name in ("on_train_batch_end", "apply_model", "sample_log", "decode_first_stage", "q_sample", "ema_scope", "get_learned_conditioning"):
        assert callable(getattr(LatentDiffusion, name)), name


def test_first_stage_config_is_built_or_refused_never_dropped():
    """A {target, params} first_stage_config (what the reference's YAML passes) must not silently become `first_stage_model = None`:
    get_input would then feed all-zero latents to a training loop."""
    import pytest
    from stedm_amd.latent_diffusion import LatentDiffusion, S_ZSS_DM, StedmHipError
    from stedm_amd.unet import UNetModel
    tiny = dict(image_size=16, in_channels=7, model_channels=32, out_channels=4, num_res_blocks=1, attention_resolutions=[32], channel_mult=[1, 2],
                num_heads=4)
    with pytest.raises((ImportError, NotImplementedError)):
        LatentDiffusion(UNetModel(**tiny), conditioning_key="hybrid", first_stage_config={"target": "no.such.module.Stage", "params": {}})
    zm = S_ZSS_DM("swin_v2_t", {"name": "none"}, {"name": "linear"}, {"data": {"patch_size": 64}}, UNetModel(**tiny), conditioning_key="hybrid",
                  image_size=16, channels=4, cond_stage_key="segmentation")
    batch = {"image": torch.zeros(1, 64, 64, 3), "segmentation": torch.zeros(1, 64, 64, 2), "style_imgs": torch.zeros(1, 1, 64, 64, 3)}
    zm.train()
    with pytest.raises(StedmHipError):
        zm.get_input(batch, "image")


def test_swin_state_dict_names_and_size():
    """torchvision's swin_v2_t: 28 351 570 parameters with the 1000-class head (its published model card); names as its state dict."""
    from stedm_amd.swin import swin_v2_t, get_model
    m = swin_v2_t()
    assert sum(p.numel() for p in m.parameters()) == 28_351_570
    sd = m.state_dict()
    for k in ("features.0.0.weight", "features.0.2.bias", "features.1.1.attn.qkv.bias", "features.1.0.attn.logit_scale", "features.1.0.attn.cpb_mlp.0.bias",
              "features.1.0.attn.cpb_mlp.2.weight", "features.1.0.attn.relative_coords_table", "features.1.0.attn.relative_position_index",
              "features.5.5.mlp.3.weight", "features.2.reduction.weight", "features.6.norm.bias", "features.7.1.norm2.weight", "norm.weight", "head.bias"):
        assert k in sd, k
    assert not any(".stochastic_depth." in k or k.startswith("permute") for k in sd)
    assert sum(p.numel() for p in get_model("swin_v2_s").parameters()) == 49_737_442
    assert sum(p.numel() for p in get_model("swin_v2_b").parameters()) == 87_930_848


def test_default_style_agg_builds_the_hip_swin_embedder():
    """conf/config_diff.yaml:16 `style_agg: linear` (and README's `mean`) with style_sampling mp: S_ZSS_DM builds swin_v2_t with the
    Linear(768, 512) head (networks/s_zss_dm.py:19-20) on the HIP-backed containers, no torchvision needed; oracle tables agree."""
    from oracle import swin as osw
    from stedm_amd.latent_diffusion import S_ZSS_DM
    from stedm_amd.swin import SwinTransformerV2
    from stedm_amd.unet import UNetModel
    tiny = dict(image_size=16, in_channels=7, model_channels=32, out_channels=4, num_res_blocks=1, attention_resolutions=[32], channel_mult=[1, 2],
                num_heads=4)
    from types import SimpleNamespace
    mp = SimpleNamespace(name="mp", num_patches=4)
    zm = S_ZSS_DM("swin_v2_t", mp, SimpleNamespace(name="linear"), {"data": {"patch_size": 512}}, UNetModel(**tiny),
                  conditioning_key="hybrid", image_size=16, channels=4, cond_stage_key="segmentation")
    emb = zm.agg_block.embedder
    assert isinstance(emb, SwinTransformerV2) and tuple(emb.head.weight.shape) == (512, 768)
    assert "agg_block.embedder.features.5.3.attn.cpb_mlp.2.weight" in zm.state_dict() and "agg_block.linear_block.1.weight" in zm.state_dict()
    at = emb.features[1][1].attn
    assert at.shift_size == [4, 4] and emb.features[1][0].attn.shift_size == [0, 0]
    assert torch.equal(at.relative_position_index.view(64, 64), osw.relative_position_index())
    assert torch.equal(at.relative_coords_table.view(-1, 2), osw.relative_coords_table())
    assert float(at.qkv.bias[96:192].abs().max()) == 0.0
    import pytest
    with pytest.raises(NotImplementedError):
        S_ZSS_DM("resnet50", mp, SimpleNamespace(name="mean"), {"data": {"patch_size": 512}}, UNetModel(**tiny),
                 conditioning_key="hybrid", image_size=16, channels=4, cond_stage_key="segmentation")


def test_oracle_swin_window_attention_matches_per_window_bruteforce():
    """The oracle's roll / reshape / mask form against a literal per-token loop over one shifted, padded map."""
    import math
    import torch.nn.functional as F
    from oracle import swin as osw
    from stedm_amd.swin import ShiftedWindowAttentionV2
    torch.manual_seed(0)
    C, heads, H, W, shift = 64, 2, 12, 20, 4
    at = ShiftedWindowAttentionV2(C, [8, 8], [shift, shift], heads)
    p = {k: v.clone() for k, v in at.state_dict().items()}
    p["qkv.weight"] = torch.randn(3 * C, C) / 8
    p["qkv.bias"] = torch.randn(3 * C) * 0.1
    p["proj.weight"], p["proj.bias"] = torch.eye(C), torch.zeros(C)
    p["cpb_mlp.0.weight"], p["cpb_mlp.0.bias"], p["cpb_mlp.2.weight"] = torch.randn(512, 2), torch.randn(512) * 0.1, torch.randn(heads, 512) / 22
    x = torch.randn(1, H, W, C)
    want = osw.window_attention(x, p, "", heads, shift)
    pH, pW = 16, 24
    bz = p["qkv.bias"].clone(); bz[C:2 * C] = 0
    xp = F.pad(x, (0, 0, 0, pW - W, 0, pH - H))
    qkv = F.linear(xp, p["qkv.weight"], bz)[0]                       # [pH, pW, 3C], pad rows = bias
    bias = osw.position_bias(p, "", heads)
    scale = torch.clamp(p["logit_scale"], max=math.log(100.0)).exp().reshape(-1)
    region = lambda v, n: 0 if v < n - 8 else (1 if v < n - shift else 2)
    got = torch.zeros(H, W, C)
    for wy in range(pH // 8):
        for wx in range(pW // 8):
            toks = [((wy * 8 + i // 8), (wx * 8 + i % 8)) for i in range(64)]            # rolled-frame positions
            src = [((ys + shift) % pH, (xs + shift) % pW) for ys, xs in toks]
            ids = [region(ys, pH) * 3 + region(xs, pW) for ys, xs in toks]
            for hd in range(heads):
                sl = slice(hd * 32, hd * 32 + 32)
                q = torch.stack([qkv[y, x_, :C][sl] for y, x_ in src]); k = torch.stack([qkv[y, x_, C:2 * C][sl] for y, x_ in src])
                v = torch.stack([qkv[y, x_, 2 * C:][sl] for y, x_ in src])
                a = F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).T * scale[hd] + bias[hd]
                m = torch.tensor([[0.0 if ids[i] == ids[j] else -100.0 for j in range(64)] for i in range(64)])
                o = F.softmax(a + m, dim=-1) @ v
                for i, (y, x_) in enumerate(src):
                    if y < H and x_ < W:
                        got[y, x_, sl] = o[i]
    assert float((got - want[0]).abs().max()) < 1e-5


def test_batch_prefetcher_interface_without_gpu():
    """BatchPrefetcher is GPU plumbing (side stream + event): constructing it without a GPU fails loudly rather than copying on the host."""
    import pytest
    from stedm_amd.parallel import BatchPrefetcher
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_sampler.py::test_batch_prefetcher_overlaps_and_preserves_data")
    with pytest.raises(Exception):
        BatchPrefetcher("cuda:0")
