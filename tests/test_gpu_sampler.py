"""GPU tier (-m gpu): DDIM + rescaled-CFG sampling loop (reference module surface: LatentDiffusion.sample_log ->
DDIMSampler.sample) against the CPU oracle loop on identical x_T / conditioning / injected noise."""
import numpy as np
import pytest
import torch

from stedm_amd.utils import prng

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

TINY = dict(image_size=16, in_channels=7, model_channels=32, out_channels=4, num_res_blocks=2,
            attention_resolutions=[32, 16, 8], channel_mult=[1, 2, 4], num_heads=4)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def make(dev, use_graph=False, precision="parity"):
    from stedm_amd.latent_diffusion import LatentDiffusion
    from stedm_amd.unet import UNetModel
    unet = UNetModel(precision=precision, **TINY).eval()
    prng.fill_module_(unet, seed=6)
    ld = LatentDiffusion(unet, linear_start=0.0015, linear_end=0.0205, image_size=16, channels=4, conditioning_key="hybrid",
                         loss_type="l1", use_graph=use_graph)
    return ld.to(dev)


def oracle_sample(xT, cc, ctx, ctx_u, S, eta, scale, noises=None):
    from oracle import ddim as od
    from oracle import unet as ou
    cfg = ou.UNetConfig(image_size=16, in_channels=7, model_channels=32, out_channels=4, channel_mult=(1, 2, 4), num_heads=4)
    plan = ou.build_plan(cfg)
    P = prng.fill_state_dict(plan.shapes, 6)

    def apply_model(x, t, c):
        return ou.unet_forward(P, cfg, torch.cat([x, c["c_concat"][0]], 1), t, c["c_crossattn"][0], plan=plan)

    cond = {"c_concat": [cc], "c_crossattn": [ctx]}
    unc = None if ctx_u is None else {"c_concat": [cc], "c_crossattn": [ctx_u]}
    return od.ddim_sample(apply_model, od.Schedule(), xT, cond, S, eta, uncond=unc, scale=scale, noises=noises)


def inputs(B=2):
    xT = prng.normal(30, "s.xT", (B, 4, 16, 16))
    cc = prng.normal(30, "s.cc", (B, 3, 16, 16)) * 0.5
    ctx = prng.normal(30, "s.ctx", (B, 128))
    ctx_u = prng.normal(30, "s.ctxu", (B, 128))
    return xT, cc, ctx, ctx_u


def rel(a, b):
    return float((a.double().cpu() - b.double()).abs().max() / b.double().std())


@pytest.mark.parametrize("use_graph", [False, True])
def test_ddim_cfg_loop_vs_oracle(dev, use_graph):
    xT, cc, ctx, ctx_u = inputs()
    ref = oracle_sample(xT, cc, ctx, ctx_u, 5, 0.0, 1.5)
    ld = make(dev, use_graph)
    cond = {"c_concat": [cc.to(dev)], "c_crossattn": [ctx.to(dev)]}
    unc = {"c_concat": [cc.to(dev).clone()], "c_crossattn": [ctx_u.to(dev)]}   # equal content, different storage
    s, inter = ld.sample_log(cond, 2, True, 5, eta=0.0, x_T=xT.to(dev), unconditional_conditioning=unc,
                             unconditional_guidance_scale=1.5, log_every_t=1000)
    err = rel(s, ref)
    print(f"[ddim cfg x5, graph={use_graph}] rel err vs oracle loop: {err:.3e}")
    assert err < 1e-3
    assert len(inter["x_inter"]) == 3   # initial, index 4 (== total-1), index 0


@pytest.mark.parametrize("size", [24, 40])
def test_ddim_cfg_loop_on_a_latent_size_off_the_tile_grid_vs_oracle(dev, size):
    """The whole sampling loop (shared-encoder CFG pass, fused DDIM update, hipGraph replay) on latents whose width is not a power of two: the
    3x3 convolutions run as im2col + flat GEMM (stedm_im2col_rows16), the attention sees 36 / 100 tokens. Final latents vs the CPU oracle loop."""
    from oracle import ddim as od
    from oracle import unet as ou
    from stedm_amd.latent_diffusion import LatentDiffusion
    from stedm_amd.unet import UNetModel
    kw = dict(TINY, image_size=size, num_res_blocks=1)
    unet = UNetModel(precision="parity", **kw).eval()
    prng.fill_module_(unet, seed=9)
    ld = LatentDiffusion(unet, linear_start=0.0015, linear_end=0.0205, image_size=size, channels=4, conditioning_key="hybrid", loss_type="l1",
                         use_graph=True).to(dev)
    cfg = ou.UNetConfig(image_size=size, in_channels=7, model_channels=32, out_channels=4, num_res_blocks=1, channel_mult=(1, 2, 4), num_heads=4)
    plan = ou.build_plan(cfg)
    P = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
    B = 2
    xT = prng.normal(32, "sg.xT", (B, 4, size, size)); cc = prng.normal(32, "sg.cc", (B, 3, size, size)) * 0.5
    ctx = prng.normal(32, "sg.ctx", (B, 128)); ctx_u = prng.normal(32, "sg.ctxu", (B, 128))
    apply_model = lambda x, t, c: ou.unet_forward(P, cfg, torch.cat([x, c["c_concat"][0]], 1), t, c["c_crossattn"][0], plan=plan)
    ref = od.ddim_sample(apply_model, od.Schedule(), xT, {"c_concat": [cc], "c_crossattn": [ctx]}, 4, 0.0,
                         uncond={"c_concat": [cc], "c_crossattn": [ctx_u]}, scale=1.5)
    cond = {"c_concat": [cc.to(dev)], "c_crossattn": [ctx.to(dev)]}
    unc = {"c_concat": [cc.to(dev)], "c_crossattn": [ctx_u.to(dev)]}
    for mode, tol in (("parity", 1e-3), ("f16", 1e-2)):
        unet.set_precision(mode)
        s, _ = ld.sample_log(cond, B, True, 4, eta=0.0, x_T=xT.to(dev), unconditional_conditioning=unc, unconditional_guidance_scale=1.5, log_every_t=1000)
        err = rel(s, ref)
        print(f"[ddim cfg x4 on {size}x{size} latents, graph replay, {mode}] rel err vs oracle loop: {err:.3e}")
        assert err < tol, mode


def test_graph_equals_eager_bits(dev):
    xT, cc, ctx, ctx_u = inputs()
    outs = []
    for g in (False, True):
        ld = make(dev, g, "f16")
        cond = {"c_concat": [cc.to(dev)], "c_crossattn": [ctx.to(dev)]}
        unc = {"c_concat": [cc.to(dev)], "c_crossattn": [ctx_u.to(dev)]}
        s, _ = ld.sample_log(cond, 2, True, 4, eta=0.0, x_T=xT.to(dev), unconditional_conditioning=unc,
                             unconditional_guidance_scale=1.5)
        outs.append(s.clone())
    assert torch.equal(outs[0], outs[1])


def test_ddim_eta1_injected_noise_and_no_cfg(dev):
    xT, cc, ctx, _ = inputs()
    S = 4
    noises = [prng.normal(31, f"nz{i}", (2, 4, 16, 16)) for i in range(S)]
    ref = oracle_sample(xT, cc, ctx, None, S, 1.0, 1.0, noises=noises)
    ld = make(dev)
    cond = {"c_concat": [cc.to(dev)], "c_crossattn": [ctx.to(dev)]}
    s, _ = ld.sample_log(cond, 2, True, S, eta=1.0, x_T=xT.to(dev), noises=noises)
    assert rel(s, ref) < 1e-3


def test_cfg_pass_equals_two_sequential_forwards(dev):
    """The shared-encoder CFG pass must give the same values as the reference's two sequential apply_model calls."""
    xT, cc, ctx, ctx_u = inputs(3)
    ld = make(dev, precision="f16")
    t = torch.tensor([951, 951, 951], dtype=torch.long, device=dev)
    cond = {"c_concat": [cc.to(dev)], "c_crossattn": [ctx.to(dev)]}
    unc = {"c_concat": [cc.to(dev)], "c_crossattn": [ctx_u.to(dev)]}
    e_c = ld.apply_model(xT.to(dev), t, cond).clone()
    e_u = ld.apply_model(xT.to(dev), t, unc).clone()
    f_c, f_u = ld.apply_model_cfg(xT.to(dev), t, cond, unc)
    assert torch.equal(e_c, f_c) and torch.equal(e_u, f_u)
    # different c_concat -> falls back to two passes
    unc2 = {"c_concat": [cc.to(dev) * 0.5], "c_crossattn": [ctx_u.to(dev)]}
    g_c, g_u = ld.apply_model_cfg(xT.to(dev), t, cond, unc2)
    assert torch.equal(g_c, e_c) and not torch.equal(g_u, e_u)


def test_predict_step_surface_end_to_end(dev):
    """S_ZSS_DM.get_input + predict_step call sequence (ldm_diffusion.py:76-91) with the HIP sViT / SpatialRescaler / U-Net /
    DDIM sampler, against the same pipeline assembled from the CPU oracle pieces."""
    from oracle import ddim as od
    from oracle import style as ost
    from oracle import unet as ou
    from stedm_amd.latent_diffusion import S_ZSS_DM, predict_latents
    from stedm_amd.unet import UNetModel
    B, P, ns = 2, 64, 4
    ucfg = dict(image_size=16, in_channels=7, model_channels=128, out_channels=4, num_res_blocks=1,
                attention_resolutions=[32, 16, 8], channel_mult=[1, 2], num_heads=4)
    unet = UNetModel(**ucfg).eval()
    prng.fill_module_(unet, seed=50)
    agg = dict(name="svit", patch_size=8, dim=256, depth=2, heads=12, mlp_dim=256, pool="mean", channels=3, dropout=0.1,
               emb_dropout=0.1, t_dim=256)
    model = S_ZSS_DM("swin_v2_t", dict(name="mp", num_patches=ns), agg, {"data": {"patch_size": P}}, unet,
                     linear_start=0.0015, linear_end=0.0205, image_size=16, channels=4, conditioning_key="hybrid", loss_type="l1",
                     cond_stage_key="segmentation", cond_stage_config={"target": "ldm.modules.encoders.modules.SpatialRescaler",
                                                                      "params": {"n_stages": 2, "in_channels": 2, "out_channels": 3}})
    prng.fill_module_(model.agg_block, seed=51)
    prng.fill_module_(model.cond_stage_model, seed=52)
    model = model.to(dev).eval()
    img = prng.uniform(53, "ps.img", (B, P, P, 3))
    seg = (prng.uniform(53, "ps.seg", (B, P, P, 2)) > 0).float()
    sty = prng.uniform(53, "ps.sty", (B, ns, P, P, 3))
    xT = prng.normal(53, "ps.xT", (B, 4, 16, 16))
    batch = {"image": img.to(dev), "segmentation": seg.to(dev), "style_imgs": sty.to(dev)}
    got = predict_latents(model, batch, ddim_steps=4, eta=0.0, cfg_scale=1.5, x_T=xT.to(dev))
    # ---- oracle pipeline
    ocfg = ou.UNetConfig(image_size=16, in_channels=7, model_channels=128, out_channels=4, num_res_blocks=1, channel_mult=(1, 2), num_heads=4)
    plan = ou.build_plan(ocfg)
    PU = prng.fill_state_dict(plan.shapes, 50)
    scfg = ost.SViTConfig(image_size=P, ns=ns, depth=2)
    PS = prng.fill_state_dict(ost.svit_shapes(scfg), 51)
    for l in range(2):
        PS[f"transformer.layers.{l}.0.fn.temperature"] = torch.tensor(float(np.log(64 ** -0.5)))
    wmap = prng.fill_value(52, "channel_mapper.weight", (3, 2, 1, 1))
    cc = ost.spatial_rescaler(seg.permute(0, 3, 1, 2), wmap)
    ctx = ost.svit_forward(PS, scfg, sty)
    ctx_u = ost.svit_forward(PS, scfg, torch.zeros_like(sty) - 2)
    am = lambda x, t, c: ou.unet_forward(PU, ocfg, torch.cat([x, c["c_concat"][0]], 1), t, c["c_crossattn"][0], plan=plan)
    ref = od.ddim_sample(am, od.Schedule(), xT, {"c_concat": [cc], "c_crossattn": [ctx]}, 4, 0.0,
                         uncond={"c_concat": [cc], "c_crossattn": [ctx_u]}, scale=1.5)
    err = rel(got, ref)
    print(f"[predict_step surface] rel err vs oracle pipeline: {err:.3e}")
    assert err < 1e-3
    # the reference's literal second get_input over the constant batch gives the same latents as the one-sample broadcast
    lit = predict_latents(model, batch, ddim_steps=4, eta=0.0, cfg_scale=1.5, x_T=xT.to(dev), dedup_uncond=False)
    assert rel(lit, ref) < 1e-3 and rel(lit, got.cpu()) < 1e-4


def test_bench_step_graph_replay_equals_eager_and_tracks_parity_mode(dev):
    """The exact step bench.py times (NS32, 64 latents, DDIM-50 + CFG 1.5, bf16 single product, hipGraph replay): replayed steps
    give the bits of eager steps, and three denoising steps stay within the single-product budget of the parity mode."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_model, synth_inputs
    from stedm_amd.ddim import DDIMSampler, StepGraph
    B, nsteps = 64, 3
    xT, cond, unc = synth_inputs(dev, B, 0, 1)
    finals = {}
    for precision in ("bf16", "parity"):
        ld = build_model(dev, precision)
        smp = DDIMSampler(ld, use_graph=True)
        smp.make_schedule(50, ddim_eta=0.0, verbose=False)
        n = smp.ddim_timesteps.shape[0]
        img_e = xT.clone()
        sg = StepGraph(smp, img_e, cond, unc, 1.5)
        sg.reset(n - 1)
        for _ in range(nsteps):
            sg.step_eager()
        torch.cuda.synchronize()
        finals[precision] = img_e.clone()
        if precision == "bf16":
            img_g = xT.clone()
            sg2 = StepGraph(smp, img_g, cond, unc, 1.5)
            sg2.reset(n - 1)
            sg2.step_eager()
            with sg2.stream_ctx():
                sg2.capture()
                for _ in range(nsteps - 1):
                    sg2.replay()
                torch.cuda.current_stream().synchronize()
            sg2.join()
            torch.cuda.synchronize()
            assert torch.equal(img_g, img_e)
        del ld, smp, sg
    a, b = finals["bf16"].double().cpu(), finals["parity"].double().cpu()
    l2 = float((a - b).norm() / b.norm())
    print(f"[bench step x{nsteps}] bf16 vs parity mode latents: rel-L2 {l2:.3e}")
    assert l2 < 5e-3


@pytest.mark.gpu
def test_image_epilogue_is_bit_exact(dev):
    """predict_step's uint8 conversion + segmentation argmax (ldm_diffusion.py:93-99): integer outputs, bit-exact against numpy,
    including the clip limits, values next to them, and values that land on integer boundaries."""
    from oracle import post as opost
    from stedm_amd import ops
    from stedm_amd.latent_diffusion import images_for_saving
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 3, 40, 24, generator=g) * 0.9
    flat = x.view(-1)
    k = torch.arange(256, dtype=torch.float32)
    edge = torch.cat([k / 127.5 - 1.0, torch.nextafter(k / 127.5 - 1.0, torch.tensor(2.0)), torch.nextafter(k / 127.5 - 1.0, torch.tensor(-2.0)),
                      torch.tensor([-1.0, 1.0, -1.5, 1.5, 0.0, -0.0, 0.9999999, -0.9999999])])
    flat[: edge.numel()] = edge
    seg = torch.rand(3, 40, 24, 2, generator=g)
    seg[0, 0, :4] = torch.tensor([[0.5, 0.5], [1.0, 0.0], [0.0, 1.0], [0.25, 0.25]])     # ties -> first index
    img, cls = images_for_saving(x.to(dev), seg.to(dev))
    assert img.dtype == torch.uint8 and tuple(img.shape) == (3, 40, 24, 3)
    assert np.array_equal(img.cpu().numpy(), opost.image_to_uint8(x.numpy()))
    assert np.array_equal(cls.cpu().numpy(), opost.segmentation_to_uint8(seg.numpy()))
    seg5 = torch.rand(2, 8, 8, 5, generator=g)
    assert np.array_equal(ops.argmax_u8(seg5.to(dev)).cpu().numpy(), opost.segmentation_to_uint8(seg5.numpy()))


@pytest.mark.gpu
def test_prepare_batch_matches_the_reference_lines(dev):
    """prepare_batch (ldm_diffusion.py:51-60): NHWC views + the segmentation class merge, against the reference's torch lines restated."""
    from stedm_amd.latent_diffusion import prepare_batch
    g = torch.Generator().manual_seed(8)
    img = torch.rand(3, 3, 20, 12, generator=g)
    cls = torch.randint(0, 4, (3, 20, 12), generator=g)
    seg = torch.nn.functional.one_hot(cls, 4).permute(0, 3, 1, 2).float()
    sty = torch.rand(3, 2, 3, 16, 16, generator=g)
    out = prepare_batch((img.to(dev), seg.to(dev), None, sty.to(dev)))
    ref = seg.permute(0, 2, 3, 1).clone()
    ref[:, :, :, 1] = torch.sum(ref[:, :, :, 1:], dim=-1)
    assert torch.equal(out["segmentation"].cpu(), ref[:, :, :, :2])
    assert torch.equal(out["image"].cpu(), img.permute(0, 2, 3, 1)) and torch.equal(out["style_imgs"].cpu(), sty.permute(0, 1, 3, 4, 2))


def test_batch_prefetcher_overlaps_and_preserves_data(dev):
    """the next predict batch crosses PCIe on a side stream while the compute stream works: same bytes arrive, the compute stream only waits
    for the copy's event, and two submissions in flight do not alias."""
    from stedm_amd.parallel import BatchPrefetcher
    pf = BatchPrefetcher(dev)
    g = torch.Generator().manual_seed(3)
    hosts = [{"a": torch.randn(64, 1024, generator=g).pin_memory(), "b": torch.randn(7, 3, generator=g).pin_memory(), "tag": i} for i in range(3)]
    h0 = pf.submit(hosts[0])
    h1 = pf.submit(hosts[1])
    busy = torch.randn(2048, 2048, device=dev)
    for _ in range(10):
        busy = busy @ busy * 1e-3           # work on the compute stream while the copies fly
    d0, d1 = pf.get(h0), pf.get(h1)
    h2 = pf.submit(hosts[2])
    d2 = pf.get(h2)
    torch.cuda.synchronize()
    for d, hst in zip((d0, d1, d2), hosts):
        assert d["tag"] == hst["tag"] and torch.equal(d["a"].cpu(), hst["a"]) and torch.equal(d["b"].cpu(), hst["b"])
    assert d0["a"].data_ptr() != d1["a"].data_ptr()


def test_device_per_sample_normal_matches_its_numpy_restatement_and_is_shard_invariant(dev):
    """stedm_philox_normal: x_T / per-step noise of a rank's shard drawn on the device. Against the numpy restatement
    of its definition (oracle/dropmask.py normal_rows: Philox4x32-10 words are integers and must agree exactly, the Box-Muller arithmetic
    to a few float32 ulps); N(0, 1) moments; the rows of a shard are the rows of the whole batch (any world size gives the same samples);
    distinct streams and seeds give distinct rows."""
    from oracle import dropmask as od
    from stedm_amd import parallel as par
    shape = (4, 32, 32)
    n = 4 * 32 * 32
    full = par.per_sample_normal_device(1234, 0, 10, shape, 3, dev)
    ref = torch.from_numpy(od.normal_rows(1234, range(10), n, 3)).view(10, *shape)
    d = (full.cpu() - ref).abs().max()
    print(f"[device normal] max |device - numpy restatement| = {float(d):.2e}; mean {float(full.mean()):+.4f}, std {float(full.std()):.4f}")
    assert float(d) < 2e-5
    assert abs(float(full.mean())) < 0.02 and abs(float(full.std()) - 1.0) < 0.02
    for world in (2, 3, 8):
        parts = [par.per_sample_normal_device(1234, *(lambda lo, hi: (lo, hi - lo))(*par.shard_range(10, r, world)), shape, 3, dev) for r in range(world)]
        assert torch.equal(torch.cat(parts), full)
    assert not torch.equal(par.per_sample_normal_device(1234, 0, 10, shape, 4, dev), full)
    assert not torch.equal(par.per_sample_normal_device(1235, 0, 10, shape, 3, dev), full)
    odd = par.per_sample_normal_device(7, 5, 3, (13,), 0, dev)          # a row length that is not a multiple of 4
    ref2 = torch.from_numpy(od.normal_rows(7, range(5, 8), 13, 0))
    assert float((odd.cpu() - ref2).abs().max()) < 2e-5
