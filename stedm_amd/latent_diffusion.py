"""Model surface that `modules/ldm_diffusion.py::LDM_Diffusion` drives (SURVEY.md §8b seam 2), HIP-backed.

  DiffusionWrapper   ddpm.py:1398-1424  (hybrid conditioning: cat([x]+c_concat,1), cat(c_crossattn,1))
  LatentDiffusion    ddpm.py:424-…     subset on the hot path: register_schedule :120-172, q_sample :277-280,
                                        apply_model :894-995 (live lines), p_losses :1015-1048 (forward value),
                                        sample_log :1237-1250, get_learned_conditioning :554-565
The first stage (VQ-f4 autoencoder) is outside this round's scope: `first_stage_model`
is an optional caller-supplied module; `decode_first_stage` raises if it is absent.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .ddim import DDIMSampler
from .schedule import NoiseSchedule
from .unet import UNetModel


def instantiate_from_config(config):
    """ldm/util.py:78-93 restricted to targets this package provides (reference target strings are mapped)."""
    import importlib
    if "target" not in config:
        raise KeyError("Expected key `target` to instantiate.")
    alias = {
        "ldm.modules.diffusionmodules.openaimodel.UNetModel": "stedm_amd.unet.UNetModel",
        "ldm.modules.encoders.modules.SpatialRescaler": "stedm_amd.style.SpatialRescaler",
    }
    target = alias.get(config["target"], config["target"])
    module, cls = target.rsplit(".", 1)
    return getattr(importlib.import_module(module), cls)(**config.get("params", dict()))


class DiffusionWrapper(nn.Module):
    """ddpm.py:1398-1424."""

    def __init__(self, diff_model_config, conditioning_key):
        super().__init__()
        self.diffusion_model = diff_model_config if isinstance(diff_model_config, nn.Module) else instantiate_from_config(diff_model_config)
        self.conditioning_key = conditioning_key
        assert self.conditioning_key in [None, 'concat', 'crossattn', 'hybrid', 'adm']
        if self.conditioning_key != 'hybrid':
            raise NotImplementedError("only conditioning_key='hybrid' is used by STEDM (conf/diffusion/ldm_based.yaml:12)")

    @torch.no_grad()
    def forward(self, x, t, c_concat: list = None, c_crossattn: list = None, out=None, uniform_t=False):
        cc = c_crossattn[0] if len(c_crossattn) == 1 else torch.cat(c_crossattn, 1)
        xc = c_concat[0] if len(c_concat) == 1 else torch.cat(c_concat, 1)
        return self.diffusion_model.forward_parts(x, xc, t, cc, out=out, uniform_t=uniform_t)   # cat folded into the first conv

    def _same_tensor(self, a, b) -> bool:
        """Content equality of two conditioning tensors, decided once per (storage, version) pair (no per-step sync). An entry
        keeps both tensors alive, so a freed tensor's address cannot reappear under a stale verdict."""
        if a is b or (a.shape == b.shape and a.data_ptr() == b.data_ptr()):
            return True
        if a.shape != b.shape:
            return False
        key = (a.data_ptr(), a._version, b.data_ptr(), b._version, tuple(a.shape))
        cache = self.__dict__.setdefault("_eq_cache", {})
        if key not in cache:
            if len(cache) > 64:
                cache.clear()
            cache[key] = (bool(torch.equal(a, b)), a, b)
        return cache[key][0]

    @torch.no_grad()
    def forward_cfg(self, x, t, cond: dict, uncond: dict, out=None, uniform_t=False):
        """cond/uncond evaluations of ddim.py:177-178 in one shared-encoder pass. Requires equal c_concat (the
        reference's unconditional batch keeps the segmentation, ldm_diffusion.py:86); otherwise two passes."""
        cc_c, cc_u = cond["c_concat"], uncond["c_concat"]
        same = len(cc_c) == len(cc_u) and all(self._same_tensor(a, b) for a, b in zip(cc_c, cc_u))
        if not same:
            B = x.shape[0]
            if out is None:
                out = torch.empty((2 * B, self.diffusion_model.out_channels) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
            self.forward(x, t, **cond, out=out[:B], uniform_t=uniform_t)
            self.forward(x, t, **uncond, out=out[B:], uniform_t=uniform_t)
            return out[:B], out[B:]
        xc = cc_c[0] if len(cc_c) == 1 else torch.cat(cc_c, 1)
        ca = cond["c_crossattn"]; cu = uncond["c_crossattn"]
        ca = ca[0] if len(ca) == 1 else torch.cat(ca, 1)
        cu = cu[0] if len(cu) == 1 else torch.cat(cu, 1)
        return self.diffusion_model.forward_cfg(x, xc, t, ca, cu, out=out, uniform_t=uniform_t)


class LatentDiffusion(nn.Module):
    """Hot-path subset of ddpm.py::LatentDiffusion / DDPM with the attribute names its callers read."""

    def __init__(self, unet_config, timesteps=1000, beta_schedule="linear", linear_start=1e-4, linear_end=2e-2,
                 loss_type="l2", image_size=256, channels=3, conditioning_key=None, parameterization="eps",
                 cond_stage_config=None, first_stage_config=None, cond_stage_trainable=False, first_stage_key="image",
                 cond_stage_key="image", log_every_t=100, scale_factor=1.0, use_graph=False, **ignored):
        super().__init__()
        assert parameterization == "eps", "the reference configs use eps-prediction"
        self.parameterization = parameterization
        self.image_size = image_size
        self.channels = channels
        self.first_stage_key = first_stage_key
        self.cond_stage_key = cond_stage_key
        self.cond_stage_trainable = cond_stage_trainable
        self.log_every_t = log_every_t
        self.loss_type = loss_type
        self.scale_factor = scale_factor
        self.learn_logvar = False
        self.use_graph = use_graph
        self.model = DiffusionWrapper(unet_config, conditioning_key)
        self.cond_stage_model = None
        if cond_stage_config is not None:
            self.cond_stage_model = cond_stage_config if isinstance(cond_stage_config, nn.Module) else instantiate_from_config(cond_stage_config)
        self.first_stage_model = first_stage_config if isinstance(first_stage_config, nn.Module) else None
        self.register_schedule(beta_schedule=beta_schedule, timesteps=timesteps, linear_start=linear_start, linear_end=linear_end)
        self.register_buffer("logvar", torch.zeros(self.num_timesteps))

    # ------------------------------------------------------------------------------------------ schedule
    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4,
                          linear_end=2e-2, cosine_s=8e-3):
        """ddpm.py:120-172 (the buffers the hot path reads)."""
        assert given_betas is None
        ns = NoiseSchedule.make(timesteps, linear_start, linear_end, beta_schedule)
        self.num_timesteps = ns.num_timesteps
        self.linear_start, self.linear_end = linear_start, linear_end
        for name in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod"):
            self.register_buffer(name, torch.from_numpy(getattr(ns, name).copy()))

    @property
    def device(self):
        return self.betas.device

    # ------------------------------------------------------------------------------------------ conditioning
    def get_learned_conditioning(self, c):
        """ddpm.py:554-565 with a module cond stage (SpatialRescaler has `encode`)."""
        if hasattr(self.cond_stage_model, 'encode') and callable(self.cond_stage_model.encode):
            return self.cond_stage_model.encode(c)
        return self.cond_stage_model(c)

    # ------------------------------------------------------------------------------------------ denoiser
    @staticmethod
    def _as_cond_dict(cond, key='c_crossattn'):
        if isinstance(cond, dict):
            return cond
        if not isinstance(cond, list):
            cond = [cond]
        return {key: cond}

    @torch.no_grad()
    def apply_model(self, x_noisy, t, cond, return_ids=False, out=None, uniform_t=False):
        """ddpm.py:894-903 + 989-995. uniform_t: all entries of t are equal (the DDIM loop, ddim.py:141)."""
        return self.model(x_noisy, t, **self._as_cond_dict(cond), out=out, uniform_t=uniform_t)

    @torch.no_grad()
    def apply_model_cfg(self, x_noisy, t, cond, uncond, out=None, uniform_t=False):
        """(e_t, e_t_uncond) of ddim.py:177-178 in one pass (see UNetModel.forward_cfg)."""
        return self.model.forward_cfg(x_noisy, t, self._as_cond_dict(cond), self._as_cond_dict(uncond), out=out, uniform_t=uniform_t)

    # ------------------------------------------------------------------------------------------ training-side forward values
    @torch.no_grad()
    def q_sample(self, x_start, t, noise=None):
        """ddpm.py:277-280 (+ extract_into_tensor util.py:96-99): one elementwise HIP kernel, per-sample scalars gathered from the
        fp32 schedule buffers by t."""
        noise = torch.randn_like(x_start) if noise is None else noise
        return ops.q_sample(x_start.float().contiguous(), noise.float().contiguous(), t.to(torch.int64).contiguous(),
                            self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod)

    @torch.no_grad()
    def p_losses(self, x_start, cond, t, noise=None):
        """ddpm.py:1015-1048, forward value only (loss_type l1 on the HIP loss kernel, l2 reported through torch for completeness;
        logvar == 0). For the loss WITH gradients see p_losses_backward."""
        noise = torch.randn_like(x_start) if noise is None else noise
        x_noisy = self.q_sample(x_start, t, noise)
        model_output = self.apply_model(x_noisy, t, cond)
        if self.loss_type == 'l1':
            loss = ops.l1_loss(model_output.contiguous(), noise.float().contiguous(), None,
                               torch.empty((1024,), dtype=torch.float64, device=x_noisy.device),
                               torch.empty((1,), dtype=torch.float32, device=x_noisy.device))[0]
        else:
            loss = ((noise - model_output) ** 2).mean([1, 2, 3]).mean()
        return loss, {"val/loss_simple": loss, "val/loss": loss}

    # ------------------------------------------------------------------------------------------ training step (U-Net parameters)
    def configure_trainer(self, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                          ema_decay: Optional[float] = 0.9999):
        """AdamW over the U-Net's parameters (modules/ldm_diffusion.py:224-234 builds `torch.optim.AdamW(self._model.model.parameters()
        ...)`) + the EMA shadow of ddpm.py:369-371 / ema.py:25-44, as one fused kernel (stedm_amd/train.py)."""
        from .train import UNetTrainer
        # the reference's parameter list: model.model (the U-Net) + cond_stage_model when cond_stage_trainable (true in
        # conf/diffusion/ldm_based.yaml:13 at configure time); the aggregation block is NOT in it (`hasattr(self.model, "embedder")` is
        # False, ldm_diffusion.py:230) — the style encoder is not trained by the reference as wired
        extra = []
        if self.cond_stage_trainable and self.cond_stage_model is not None and hasattr(self.cond_stage_model, "backward"):
            extra = [p for p in self.cond_stage_model.parameters() if p.requires_grad]
        self._trainer = UNetTrainer(self.model.diffusion_model, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, ema_decay=ema_decay,
                                    extra_params=extra)
        return self._trainer

    @torch.no_grad()
    def p_losses_backward(self, x_start, cond, t, noise=None, cond_input=None):
        """p_losses (ddpm.py:1015-1048) followed by the backward pass autograd runs for the reference in training_step
        (ddpm.py:345-358): loss_type l1, eps-parameterisation, logvar == 0. Fills `.grad` of the U-Net's parameters and returns
        (loss, loss_dict, dL/dx_noisy over [x | c_concat], dL/dc_crossattn). With `cond_input` (the raw layout fed to the cond stage) the
        SpatialRescaler's channel-mapper gradient is filled too; the style encoder's backward is not built."""
        if self.loss_type != 'l1':
            raise NotImplementedError("the training step is built for loss_type 'l1' (conf/diffusion/ldm_based.yaml)")
        tr = getattr(self, "_trainer", None) or self.configure_trainer()
        noise = torch.randn_like(x_start) if noise is None else noise
        x_noisy = self.q_sample(x_start, t, noise)
        cd = self._as_cond_dict(cond)
        cc, ca = cd["c_concat"], cd["c_crossattn"]
        xc = cc[0] if len(cc) == 1 else torch.cat(cc, 1)
        ctx = ca[0] if len(ca) == 1 else torch.cat(ca, 1)
        if tr.extra_params and cond_input is None:
            raise ValueError("the cond stage is in the optimizer (cond_stage_trainable): pass cond_input (the raw layout given to the cond stage) so "
                             "that its gradient can be computed")
        loss, dx, dctx = tr.loss_and_backward(x_noisy, xc, t, ctx, noise)
        if cond_input is not None and hasattr(self.cond_stage_model, "backward"):
            # cond_stage_trainable (s_zss_dm.py:46-48): the layout conditioner's channel mapper receives the c_concat slice of dL/dx
            self.cond_stage_model.backward(cond_input, dx[:, x_noisy.shape[1]:].contiguous())
        return loss, {"train/loss_simple": loss, "train/loss": loss}, dx, dctx

    @torch.no_grad()
    def training_step_hip(self, x_start, cond, t=None, noise=None, cond_input=None):
        """One optimisation step on a prepared batch (latents + conditioning as `get_input` returns them): t ~ U{0..T-1}
        (ddpm.py:878), p_losses + backward, AdamW + EMA. Returns the loss (device tensor)."""
        if t is None:
            t = torch.randint(0, self.num_timesteps, (x_start.shape[0],), device=x_start.device).long()
        loss, _, _, _ = self.p_losses_backward(x_start, cond, t, noise, cond_input=cond_input)
        self._trainer.optimizer_step()
        return loss

    # ------------------------------------------------------------------------------------------ checkpoints (reference key layout)
    @staticmethod
    def _ema_name(param_name: str) -> str:
        """LitEma registers its shadow of `model.<name>` as a buffer named `<name>` without dots (ema.py:17-21)."""
        return param_name.replace('.', '')

    @torch.no_grad()
    def load_reference_state_dict(self, ckpt: dict, use_ema: bool = False):
        """Load a checkpoint written by the reference: a Lightning checkpoint ({"state_dict": ...}) of `LDM_Diffusion`, whose keys carry
        the `_model.` prefix (modules/ldm_diffusion.py:38), or a bare `LatentDiffusion.state_dict()`. U-Net, cond stage and
        aggregation block load by name (same parameter names and OIHW shapes); `model_ema.*` (ema.py) goes to the trainer's EMA
        shadows, or — with use_ema, like `ema_scope` (ddpm.py:174-188) — into the U-Net itself. `first_stage_model.*` is loaded only
        when a first stage was supplied. Returns (missing, unexpected) key lists like `load_state_dict(strict=False)`."""
        sd = ckpt.get("state_dict", ckpt)
        sd = {(k[len("_model."):] if k.startswith("_model.") else k): v for k, v in sd.items()}
        ema = {k[len("model_ema."):]: v for k, v in sd.items() if k.startswith("model_ema.")}
        rest = {k: v for k, v in sd.items() if not k.startswith("model_ema.")
                and (self.first_stage_model is not None or not k.startswith("first_stage_model."))}
        res = self.load_state_dict(rest, strict=False)
        missing, unexpected = list(res.missing_keys), list(res.unexpected_keys)
        named = dict(self.model.named_parameters())
        if use_ema and ema:
            for name, p in named.items():
                key = self._ema_name(name)
                if key in ema:
                    p.copy_(ema[key].to(p.device, p.dtype))
            self.model.diffusion_model.invalidate()
        self._ema_loaded = {name: ema[self._ema_name(name)] for name in named if self._ema_name(name) in ema}
        self._ema_num_updates = int(ema["num_updates"]) if "num_updates" in ema else 0
        tr = getattr(self, "_trainer", None)
        if tr is not None:
            tr.load_ema({n[len("diffusion_model."):]: v for n, v in self._ema_loaded.items()}, self._ema_num_updates)
        return missing, unexpected

    def reference_state_dict(self, prefix: str = "_model.") -> dict:
        """State dict in the reference's key layout (see load_reference_state_dict), including `model_ema.*` when a trainer holds
        EMA shadows."""
        out = {prefix + k: v for k, v in self.state_dict().items()}
        tr = getattr(self, "_trainer", None)
        if tr is not None and tr.ema_named() is not None:
            out[prefix + "model_ema.decay"] = torch.tensor(tr.ema_decay, dtype=torch.float32)
            out[prefix + "model_ema.num_updates"] = torch.tensor(tr.ema_updates, dtype=torch.int)
            for name, v in tr.ema_named().items():
                out[prefix + "model_ema." + self._ema_name("diffusion_model." + name)] = v
        return out

    # ------------------------------------------------------------------------------------------ sampling
    @torch.no_grad()
    def sample_log(self, cond, batch_size, ddim, ddim_steps, **kwargs):
        """ddpm.py:1237-1250."""
        if not ddim:
            raise NotImplementedError("ancestral DDPM sampling is dead code for the shipped configs (SURVEY.md §2.1 #5)")
        sampler = DDIMSampler(self, use_graph=self.use_graph)
        shape = (self.channels, self.image_size, self.image_size)
        return sampler.sample(ddim_steps, batch_size, shape, cond, verbose=False, **kwargs)

    def decode_first_stage(self, z, **kw):
        if self.first_stage_model is None:
            raise NotImplementedError("first stage (VQ-f4 decoder) is out of this round's scope (SURVEY.md §8f next-1); "
                                      "pass a first_stage module to LatentDiffusion to use decode_first_stage")
        return self.first_stage_model.decode(z / self.scale_factor)


class S_ZSS_DM(LatentDiffusion):
    """networks/s_zss_dm.py:11-60 — LatentDiffusion + style aggregation block; `get_input` emits the hybrid conditioning
    dict {"c_concat": [layout], "c_crossattn": [style]}.

    `encoder` names the torchvision embedder of the mean/max/linear aggregators in the reference ("swin_v2_t", third-party,
    SURVEY.md §8c); here it must be supplied as a module through `embedder=` for those modes. `style_agg: svit` and
    `style_sampling: none` need no embedder."""

    def __init__(self, encoder, sampling_cfg, agg_cfg, cfg, *args, embedder: Optional[nn.Module] = None, **kwargs):
        super().__init__(*args, **kwargs)
        from . import style as st
        self._sampling_cfg = sampling_cfg
        self._agg_cfg = agg_cfg
        self._cfg = cfg
        self.embed_key = "style_imgs"
        name = lambda c: c["name"] if isinstance(c, dict) else c.name
        get = lambda c, k: c[k] if isinstance(c, dict) else getattr(c, k)
        if name(sampling_cfg) == "none":
            self._agg_block = st.Agg_None(sampling_cfg, embedder)
        elif name(agg_cfg) == "svit":
            a = dict(agg_cfg) if isinstance(agg_cfg, dict) else {k: getattr(agg_cfg, k) for k in
                                                               ("patch_size", "dim", "depth", "heads", "mlp_dim", "pool", "channels",
                                                                "dropout", "emb_dropout", "t_dim")}
            a.pop("name", None)
            data = cfg["data"] if isinstance(cfg, dict) else cfg.data
            img = data["patch_size"] if isinstance(data, dict) else data.patch_size
            ns = get(sampling_cfg, "num_patches") if name(sampling_cfg) == "mp" else 1
            self._agg_block = st.sViT(image_size=img, num_classes=512, ns=ns, **a)
        else:
            if embedder is None:
                raise NotImplementedError(f"style_agg={name(agg_cfg)!r} needs the torchvision {encoder!r} embedder (third-party, not "
                                          "part of this package): pass it as S_ZSS_DM(..., embedder=module)")
            cls = {"linear": st.Agg_Linear, "max": st.Agg_Max, "mean": st.Agg_Mean}.get(name(agg_cfg))
            if cls is None:
                raise Exception("Unkown aggregation function!")
            self._agg_block = cls(sampling_cfg, embedder)
        self.register_module("agg_block", self._agg_block)

    @torch.no_grad()
    def get_input(self, batch, k, cond_key=None, bs=None, **kwargs):
        """s_zss_dm.py:45-60 + ddpm.py:656-706. batch tensors are NHWC (LDM_Diffusion.prepare_batch, ldm_diffusion.py:51-60).
        Returns [z, {"c_concat": [c], "c_crossattn": [style]}]. Without a first stage, z is a zero placeholder of the latent
        shape (predict_step only uses len(z), ldm_diffusion.py:79-90)."""
        x = batch[k]
        if bs is not None:
            x = x[:bs]
        x = x.permute(0, 3, 1, 2).float()
        dev = self.device
        if self.first_stage_model is not None:
            z = self.first_stage_model.encode(x.to(dev)) * self.scale_factor
        else:
            z = torch.zeros((x.shape[0], self.channels, self.image_size, self.image_size), device=dev)
        xc = batch[cond_key or self.cond_stage_key]
        if bs is not None:
            xc = xc[:bs]
        xc = xc.permute(0, 3, 1, 2).float().contiguous().to(dev)
        self.cond_stage_trainable = True
        c = self.get_learned_conditioning(xc)
        self.cond_stage_trainable = False
        style_imgs = batch[self.embed_key]
        if bs is not None:
            style_imgs = style_imgs[:bs]
        style_features = self._agg_block(style_imgs.to(dev))
        return [z, {"c_concat": [c], "c_crossattn": [style_features]}]


@torch.no_grad()
def prepare_batch(batch, device=None) -> dict:
    """LDM_Diffusion.prepare_batch (modules/ldm_diffusion.py:51-60): the DataModule's tuple (image [B,3,H,W], one-hot segmentation
    [B,K,H,W], _, style images [B,n,3,H,W], ...) -> the NHWC dict `get_input` reads. The class merge of the segmentation (channel 1 =
    sum of the classes >= 1) and its NHWC layout come from one HIP kernel; image and style stack are views, as in the reference."""
    dev = device or batch[1].device
    seg = ops.seg_merge(batch[1].to(dev).float().contiguous())
    return {"image": batch[0].permute(0, 2, 3, 1), "segmentation": seg, "style_imgs": batch[3].permute(0, 1, 3, 4, 2)}


@torch.no_grad()
def predict_latents(model: S_ZSS_DM, ldm_batch: dict, ddim_steps: int, eta: float = 0.0, cfg_scale: float = 1.0,
                    style_sampling: str = "nearby", x_T: Optional[torch.Tensor] = None, dedup_uncond: bool = True):
    """Lightning-free restatement of LDM_Diffusion.predict_step (modules/ldm_diffusion.py:76-91) up to the sampled latents:
    conditional get_input, unconditional batch {image: 0, segmentation: same, style_imgs: -2}, DDIM + CFG.

    dedup_uncond: the unconditional style input is the same constant (-2) image stack for every sample (ldm_diffusion.py:86) and the
    style encoder works per sample, so its output is one vector: it is computed for ONE sample and broadcast instead of running the
    encoder over the whole constant batch; the layout conditioning of the unconditional batch is the conditional one (same
    segmentation). False: the reference's literal second get_input."""
    z, c_0 = model.get_input(ldm_batch, "image")
    kw = {} if x_T is None else {"x_T": x_T}
    if cfg_scale == 1 or style_sampling == "none":
        out, _ = model.sample_log(c_0, batch_size=len(z), ddim=True, ddim_steps=ddim_steps, eta=eta, log_every_t=1000, **kw)
    else:
        if dedup_uncond:
            sty = ldm_batch["style_imgs"]
            one = torch.full((1,) + tuple(sty.shape[1:]), -2.0, dtype=torch.float32, device=model.device)
            unc_style = model._agg_block(one).expand(len(z), -1).contiguous()
            c_uncond = {"c_concat": c_0["c_concat"], "c_crossattn": [unc_style]}
        else:
            unc_batch = {"image": torch.zeros_like(ldm_batch["image"]), "segmentation": ldm_batch["segmentation"],
                         "style_imgs": torch.zeros_like(ldm_batch["style_imgs"]) - 2}
            z, c_uncond = model.get_input(unc_batch, "image")
        out, _ = model.sample_log(c_0, batch_size=len(z), ddim=True, ddim_steps=ddim_steps, eta=eta, log_every_t=1000,
                                  unconditional_conditioning=c_uncond, unconditional_guidance_scale=cfg_scale, **kw)
    return out


@torch.no_grad()
def images_for_saving(decoded: torch.Tensor, segmentation_nhwc: Optional[torch.Tensor] = None):
    """The array work of predict_step after decode_first_stage (modules/ldm_diffusion.py:93-99): decoded [B,3,H,W] fp32 -> uint8
    [B,H,W,3] (clip to [-1,1], scale, truncate), segmentation [B,H,W,ncls] -> uint8 class map. Returned on the device; PNG
    encoding stays with the caller."""
    img = ops.image_to_uint8(decoded.float().contiguous())
    seg = None if segmentation_nhwc is None else ops.argmax_u8(segmentation_nhwc.float().contiguous())
    return img, seg
