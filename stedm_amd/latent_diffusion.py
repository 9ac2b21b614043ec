"""Model surface that `modules/ldm_diffusion.py::LDM_Diffusion` drives (SURVEY.md §8b seam 2), HIP-backed.

  DiffusionWrapper   ddpm.py:1398-1424  (hybrid conditioning: cat([x]+c_concat,1), cat(c_crossattn,1))
  LatentDiffusion    ddpm.py:424-…     subset on the hot path: register_schedule :120-172, q_sample :277-280,
                                        apply_model :894-995 (live lines), p_losses :1015-1048 (forward value),
                                        sample_log :1237-1250, get_learned_conditioning :554-565
Training seam (what `LDM_Diffusion.training_step` drives, modules/ldm_diffusion.py:63-73): `training_step(batch, batch_idx)` ->
`shared_step` -> `get_input` + `forward(x, c)` (t ~ U{0..T-1}) -> `p_losses`, and the `on_train_batch_start/end` hooks
(ddpm.py:345-371, 479-494, 868-882). In training mode with autograd enabled `p_losses` returns a loss tensor whose `backward()`
runs the hand-scheduled HIP backward (an autograd.Function bridge), so `loss.backward(); optimizer.step()` — Lightning's automatic
optimisation on one device, or a plain torch loop with torch.optim.AdamW — works unchanged; `training_step_hip` is the fused fast path
(HIP AdamW + EMA, gradient accumulation, bucketed all-reduce) that `stedm_amd.ldm_module.LDM_Diffusion` uses.
"""
from __future__ import annotations

from contextlib import contextmanager
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import StedmHipError
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


def instantiate_first_stage(config):
    """ddpm.py:530-535 (`instantiate_first_stage`): the reference target `ldm.models.autoencoder.VQModelInterface` maps to the HIP
    VQ-f4 stage when this package provides it; any other target must be importable as it stands."""
    cfg = dict(config)
    target = cfg.get("target")
    if target is None:
        raise KeyError("Expected key `target` to instantiate.")
    if target in ("ldm.models.autoencoder.VQModelInterface", "stedm_amd.vq.VQModelInterface"):
        try:
            from .vq import VQModelInterface
        except ImportError as e:
            raise NotImplementedError(f"first_stage_config target {target!r}: the HIP VQ-f4 first stage is not available ({e}); pass a module "
                                      "as first_stage_config or leave it None for sampling up to the latents") from e
        return VQModelInterface(**cfg.get("params", dict())).eval()
    return instantiate_from_config(cfg).eval()


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


class _HipTrainingLoss(torch.autograd.Function):
    """`loss = p_losses(...)` with a grad_fn: forward runs the HIP U-Net forward (tape kept by the trainer) and the L1 loss kernel,
    backward runs the hand-scheduled HIP backward and ACCUMULATES into every parameter's `.grad` (autograd's contract: several
    `loss.backward()` between two `zero_grad()` sum up — Lightning's accumulate_grad_batches relies on it). c_concat / c_crossattn
    receive their gradients as ordinary autograd inputs, so a conditioning module that runs under autograd upstream (e.g. a
    torchvision embedder) trains through this node. One forward may be outstanding at a time (the tape lives in the trainer)."""

    @staticmethod
    def forward(ctx, anchor, c_concat, c_crossattn, ld, x_start, t, noise, cond_input):
        tr = ld._trainer_or_default()
        with torch.no_grad():
            x_noisy = ld.q_sample(x_start, t, noise)
            pred = tr.forward(x_noisy, c_concat.float().contiguous(), t, c_crossattn.float().contiguous())
            loss = tr._buf("loss", (1,))
            dpred = tr._buf(f"dpred.{tuple(pred.shape)}", tuple(pred.shape))
            ops.l1_loss(pred, noise.float().contiguous(), dpred, tr._buf("loss.ws", (1024,), torch.float64), loss)
        tr._fwd_token = getattr(tr, "_fwd_token", 0) + 1
        ctx.ld, ctx.token, ctx.dpred, ctx.nx, ctx.cond_input = ld, tr._fwd_token, dpred, x_noisy.shape[1], cond_input
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g):
        ld = ctx.ld
        tr = ld._trainer
        if tr._fwd_token != ctx.token:
            raise RuntimeError("p_losses: a newer training forward replaced this one's tape; call backward() before the next forward")
        with torch.no_grad():
            had = tr.internal_grads()
            dx, dctx = tr.backward(ctx.dpred * g)
            cs = ld.cond_stage_model
            if tr.extra_params and ctx.cond_input is not None and hasattr(cs, "backward"):
                cs.backward(ctx.cond_input, dx[:, ctx.nx:].contiguous())
            tr.publish_grads(1.0, accumulate=had)
            tr._ema_pending = True
        dcc = dx[:, ctx.nx:].contiguous() if ctx.needs_input_grad[1] else None
        return torch.zeros_like(g), dcc, (dctx if ctx.needs_input_grad[2] else None), None, None, None, None, None


class LatentDiffusion(nn.Module):
    """Hot-path subset of ddpm.py::LatentDiffusion / DDPM with the attribute names its callers read."""

    def __init__(self, unet_config, timesteps=1000, beta_schedule="linear", linear_start=1e-4, linear_end=2e-2,
                 loss_type="l2", image_size=256, channels=3, conditioning_key=None, parameterization="eps",
                 cond_stage_config=None, first_stage_config=None, cond_stage_trainable=False, first_stage_key="image",
                 cond_stage_key="image", log_every_t=100, scale_factor=1.0, use_graph=False, use_ema=True, scale_by_std=False,
                 l_simple_weight=1.0, original_elbo_weight=0.0, learn_logvar=False, **ignored):
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
        if learn_logvar or original_elbo_weight != 0.0 or l_simple_weight != 1.0:
            raise NotImplementedError("learn_logvar / original_elbo_weight / l_simple_weight: the reference configs keep the defaults "
                                      "(logvar == 0, plain L1), nothing else is built")
        self.scale_by_std = scale_by_std
        self.scale_factor = scale_factor
        self.learn_logvar = False
        self.use_ema = use_ema          # ddpm.py:58,91-94: LitEma over `model`; the shadows live in the trainer (stedm_amd/train.py)
        self.use_scheduler = False      # ddpm.py:95: scheduler_config is absent from the reference configs
        self.restarted_from_ckpt = False
        self.use_graph = use_graph
        self.model = DiffusionWrapper(unet_config, conditioning_key)
        self.cond_stage_model = None
        if cond_stage_config is not None:
            self.cond_stage_model = cond_stage_config if isinstance(cond_stage_config, nn.Module) else instantiate_from_config(cond_stage_config)
        if first_stage_config is None or isinstance(first_stage_config, nn.Module):
            self.first_stage_model = first_stage_config
        else:
            # a {target, params} config (what conf/diffusion/first_stage_config/vq-f4.yaml holds): built here or refused — never
            # silently dropped (a dropped first stage would make get_input hand all-zero latents to the training step)
            self.first_stage_model = instantiate_first_stage(first_stage_config)
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

    def p_losses(self, x_start, cond, t, noise=None, cond_input=None):
        """ddpm.py:1015-1048 (loss_type l1 on the HIP loss kernel; logvar == 0, l_simple_weight 1, original_elbo_weight 0, so
        loss == loss_simple). In training mode with autograd enabled the returned loss carries a grad_fn (`_HipTrainingLoss`):
        `loss.backward()` runs the HIP backward, as autograd does for the reference in training_step (ddpm.py:345-358). Otherwise:
        the forward value only."""
        prefix = 'train' if self.training else 'val'
        if self.training and torch.is_grad_enabled():
            if self.loss_type != 'l1':
                raise NotImplementedError("the training step is built for loss_type 'l1' (conf/diffusion/ldm_based.yaml)")
            noise = torch.randn_like(x_start) if noise is None else noise
            cd = self._as_cond_dict(cond)
            cc, ca = cd["c_concat"], cd["c_crossattn"]
            xc = cc[0] if len(cc) == 1 else torch.cat(cc, 1)
            ctx = ca[0] if len(ca) == 1 else torch.cat(ca, 1)
            if cond_input is None:
                cond_input = self.__dict__.get("_last_cond_input")      # left by get_input: the raw layout the cond stage saw
            tr = self._trainer_or_default()
            if tr.extra_params and cond_input is None:
                raise ValueError("the cond stage is in the optimizer (cond_stage_trainable): its gradient needs the raw layout given to the "
                                 "cond stage — call get_input first or pass cond_input")
            loss = _HipTrainingLoss.apply(self._grad_anchor(x_start.device), xc, ctx, self, x_start.float().contiguous(),
                                          t.to(torch.int64).contiguous(), noise, cond_input)
            return loss, {f"{prefix}/loss_simple": loss.detach(), f"{prefix}/loss": loss.detach()}
        with torch.no_grad():
            noise = torch.randn_like(x_start) if noise is None else noise
            x_noisy = self.q_sample(x_start, t, noise)
            model_output = self.apply_model(x_noisy, t, cond)
            if self.loss_type == 'l1':
                loss = ops.l1_loss(model_output.contiguous(), noise.float().contiguous(), None,
                                   torch.empty((1024,), dtype=torch.float64, device=x_noisy.device),
                                   torch.empty((1,), dtype=torch.float32, device=x_noisy.device))[0]
            else:
                loss = ((noise - model_output) ** 2).mean([1, 2, 3]).mean()
        return loss, {f"{prefix}/loss_simple": loss, f"{prefix}/loss": loss}

    def _grad_anchor(self, device) -> torch.Tensor:
        """a scalar that requires grad, so that the bridge node is part of the autograd graph (not a Parameter: it must stay out of
        the state dict and of `parameters()`)"""
        a = self.__dict__.get("_anchor")
        if a is None or a.device != device:
            a = torch.zeros((), device=device, requires_grad=True)
            self.__dict__["_anchor"] = a
        return a

    # ------------------------------------------------------------------------------------------ the reference's training seam
    def get_input(self, batch, k, return_first_stage_outputs=False, force_c_encode=False, cond_key=None, return_original_cond=False, bs=None):
        """ddpm.py:656-706 for the paths the STEDM configs take: batch tensors NHWC (DDPM.get_input, ddpm.py:332-338) -> latents of
        the first stage (encode_first_stage, per-sample loop of ddpm.py:866 batched) and the conditioning: the raw cond input while
        `cond_stage_trainable` (the caller encodes it: forward / S_ZSS_DM.get_input), its encoding otherwise. -> [z, c]."""
        with torch.no_grad():
            x = batch[k]
            if x.dim() == 3:
                x = x[..., None]
            if bs is not None:
                x = x[:bs]
            x = x.permute(0, 3, 1, 2).float().contiguous().to(self.device)
            z = self.get_first_stage_encoding(self.encode_first_stage(x)).detach()
            c = xc = None
            if self.model.conditioning_key is not None:
                cond_key = self.cond_stage_key if cond_key is None else cond_key
                if cond_key in ('caption', 'coordinates_bbox', 'class_label'):
                    raise NotImplementedError(f"cond_key {cond_key!r}: text / box / class conditioning is unused by STEDM")
                if cond_key != self.first_stage_key:
                    xc = batch[cond_key]
                    if xc.dim() == 3:
                        xc = xc[..., None]
                    xc = xc.permute(0, 3, 1, 2).float().contiguous().to(self.device)
                else:
                    xc = x
                if bs is not None:
                    xc = xc[:bs]
                c = self.get_learned_conditioning(xc) if (not self.cond_stage_trainable or force_c_encode) else xc
        out = [z, c]
        if return_first_stage_outputs:
            out.extend([x, self.decode_first_stage(z)])
        if return_original_cond:
            out.append(xc)
        return out

    def encode_first_stage(self, x):
        """ddpm.py:828-866 (no fold/unfold tiling: `split_input_params` is never set)."""
        if self.first_stage_model is None:
            raise StedmHipError("no first stage: latents cannot be produced from images (pass first_stage_config)")
        return self.first_stage_model.encode(x)

    def get_first_stage_encoding(self, encoder_posterior):
        """ddpm.py:537-544: the VQ interface returns the latent tensor itself."""
        if not isinstance(encoder_posterior, torch.Tensor):
            raise NotImplementedError(f"encoder_posterior of type '{type(encoder_posterior)}' not yet implemented")
        return self.scale_factor * encoder_posterior

    def forward(self, x, c, *args, **kwargs):
        """ddpm.py:873-882: t ~ U{0..T-1} per sample, encode the conditioning when the cond stage is trainable, p_losses."""
        t = torch.randint(0, self.num_timesteps, (x.shape[0],), device=self.device).long()
        if self.model.conditioning_key is not None:
            assert c is not None
            if self.cond_stage_trainable:
                self.__dict__["_last_cond_input"] = c
                c = self.get_learned_conditioning(c)
        return self.p_losses(x, c, t, *args, **kwargs)

    def shared_step(self, batch, **kwargs):
        """ddpm.py:868-871."""
        x, c = self.get_input(batch, self.first_stage_key)[:2]
        return self(x, c)

    def training_step(self, batch, batch_idx):
        """ddpm.py:345-358: returns the loss tensor Lightning's automatic optimisation (or any `loss.backward()`) consumes."""
        loss, loss_dict = self.shared_step(batch)
        return loss

    @torch.no_grad()
    def on_train_batch_start(self, batch, batch_idx, dataloader_idx=-1):
        """ddpm.py:479-494 (rank-0-only std-rescale of the latents on the very first batch; inactive in the reference configs:
        `scale_by_std` is not set in conf/diffusion/ldm_based.yaml)."""
        if not (self.scale_by_std and batch_idx == 0 and not self.restarted_from_ckpt and not self.__dict__.get("_std_rescaled")):
            return
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
            return
        assert self.scale_factor == 1., 'rather not use custom rescaling and std-rescaling simultaneously'
        x = batch[self.first_stage_key].permute(0, 3, 1, 2).float().contiguous().to(self.device)
        z = self.get_first_stage_encoding(self.encode_first_stage(x)).detach()
        self.scale_factor = 1. / z.flatten().std()
        self.__dict__["_std_rescaled"] = True

    @torch.no_grad()
    def on_train_batch_end(self, *args, **kwargs):
        """ddpm.py:369-371: `self.model_ema(self.model)` after every micro-batch. The fused path (training_step_hip) has run it
        already; after a bridged `loss.backward()` it runs here, as one EMA-only kernel launch."""
        tr = self.__dict__.get("_trainer")
        if self.use_ema and tr is not None and getattr(tr, "_ema_pending", False):
            tr.ema_step()
            tr._ema_pending = False

    @torch.no_grad()
    def attach_optimizer(self, optimizer):
        """For a caller that keeps `torch.optim.AdamW` (the reference's configure_optimizers, modules/ldm_diffusion.py:224-234) on
        several ranks: DistributedDataParallel's reducer hooks never fire for the HIP backward (it is not an autograd graph over the
        parameters), so the gradient average over the ranks is done here, in a pre-step hook of the optimizer — the bucketed RCCL
        all-reduce of the published arena, once per optimizer step (what DDP's no_sync does for the accumulation window)."""
        def pre_step(opt, args, kwargs):
            tr = self.__dict__.get("_trainer")
            if tr is not None and getattr(tr, "_pub_arena", None) is not None:
                from .parallel import all_reduce_buckets
                world = all_reduce_buckets(tr._pub_arena, 256 * (1 << 20) // 4)
                if world > 1:
                    tr._pub_arena.mul_(1.0 / world)
        optimizer.register_step_pre_hook(pre_step)
        return optimizer

    def _trainer_or_default(self):
        tr = self.__dict__.get("_trainer")
        return tr if tr is not None else self.configure_trainer()

    @contextmanager
    def ema_scope(self, context=None):
        """ddpm.py:174-188: run the body with the EMA weights in the U-Net, restore the training weights afterwards."""
        tr = self.__dict__.get("_trainer")
        named = tr.ema_named() if (self.use_ema and tr is not None) else None
        unet = self.model.diffusion_model
        if named:
            with torch.no_grad():
                saved = {n: p.detach().clone() for n, p in unet.named_parameters() if n in named}
                for n, p in unet.named_parameters():
                    if n in named:
                        p.copy_(named[n])
            unet.invalidate()
        try:
            yield None
        finally:
            if named:
                with torch.no_grad():
                    for n, p in unet.named_parameters():
                        if n in saved:
                            p.copy_(saved[n])
                unet.invalidate()

    # ------------------------------------------------------------------------------------------ training step (fused fast path)
    def configure_trainer(self, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                          ema_decay: Optional[float] = 0.9999, accumulate_grad_batches: int = 1):
        """AdamW over the U-Net's parameters (modules/ldm_diffusion.py:224-234 builds `torch.optim.AdamW(self._model.model.parameters()
        ...)`) + the EMA shadow of ddpm.py:369-371 / ema.py:25-44, as one fused kernel (stedm_amd/train.py).
        accumulate_grad_batches: `Trainer(accumulate_grad_batches=4)` of train_diff.py:76. A checkpoint loaded before this call
        (load_reference_state_dict) hands its EMA shadows / update counter and its AdamW state over to the new trainer."""
        from .train import UNetTrainer
        # the reference's parameter list: model.model (the U-Net) + cond_stage_model when cond_stage_trainable (true in
        # conf/diffusion/ldm_based.yaml:13 at configure time); the aggregation block is NOT in it (`hasattr(self.model, "embedder")` is
        # False, ldm_diffusion.py:230) — the style encoder is not trained by the reference as wired
        extra = []
        if self.cond_stage_trainable and self.cond_stage_model is not None and hasattr(self.cond_stage_model, "backward"):
            extra = [p for p in self.cond_stage_model.parameters() if p.requires_grad]
        tr = UNetTrainer(self.model.diffusion_model, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                         ema_decay=ema_decay if self.use_ema else None, accumulate_grad_batches=accumulate_grad_batches, extra_params=extra)
        self.__dict__["_trainer"] = tr
        if self.__dict__.get("_ema_loaded"):
            tr.load_ema({n[len("diffusion_model."):]: v for n, v in self._ema_loaded.items()}, self._ema_num_updates)
        if self.__dict__.get("_opt_loaded") is not None:
            tr.load_optimizer_state_dict(self._opt_loaded)
        return tr

    @torch.no_grad()
    def p_losses_backward(self, x_start, cond, t, noise=None, cond_input=None):
        """p_losses (ddpm.py:1015-1048) followed by the backward pass autograd runs for the reference in training_step
        (ddpm.py:345-358): loss_type l1, eps-parameterisation, logvar == 0. Fills `.grad` of the U-Net's parameters and returns
        (loss, loss_dict, dL/dx_noisy over [x | c_concat], dL/dc_crossattn). With `cond_input` (the raw layout fed to the cond stage) the
        SpatialRescaler's channel-mapper gradient is filled too; the style encoder's backward is not built."""
        if self.loss_type != 'l1':
            raise NotImplementedError("the training step is built for loss_type 'l1' (conf/diffusion/ldm_based.yaml)")
        tr = self._trainer_or_default()
        noise = torch.randn_like(x_start) if noise is None else noise
        x_noisy = self.q_sample(x_start, t, noise)
        xc, ctx = self._split_cond(cond)
        if tr.extra_params and cond_input is None:
            raise ValueError("the cond stage is in the optimizer (cond_stage_trainable): pass cond_input (the raw layout given to the cond stage) so "
                             "that its gradient can be computed")
        loss, dx, dctx = tr.loss_and_backward(x_noisy, xc, t, ctx, noise)
        if cond_input is not None and hasattr(self.cond_stage_model, "backward"):
            # cond_stage_trainable (s_zss_dm.py:46-48): the layout conditioner's channel mapper receives the c_concat slice of dL/dx
            self.cond_stage_model.backward(cond_input, dx[:, x_noisy.shape[1]:].contiguous())
        return loss, {"train/loss_simple": loss, "train/loss": loss}, dx, dctx

    def _split_cond(self, cond):
        cd = self._as_cond_dict(cond)
        cc, ca = cd["c_concat"], cd["c_crossattn"]
        return (cc[0] if len(cc) == 1 else torch.cat(cc, 1)), (ca[0] if len(ca) == 1 else torch.cat(ca, 1))

    @torch.no_grad()
    def training_step_hip(self, x_start, cond, t=None, noise=None, cond_input=None, group=None, graph=False):
        """One micro-batch on the fused path (latents + conditioning as `get_input` returns them): t ~ U{0..T-1} (ddpm.py:878),
        p_losses + backward; every `accumulate_grad_batches`-th call averages the accumulated gradients over the ranks (bucketed
        all-reduce when torch.distributed is initialised — what DDP does for train_diff.py:75) and runs AdamW; LitEma's update runs
        after every micro-batch (ddpm.py:369-371). Returns the loss (device tensor). graph=True: the U-Net's part of the step (forward, loss,
        backward, AdamW + EMA) is captured once and replayed as one hipGraph launch (UNetTrainer.train_step_graphed) when the step is
        shape-static and self-contained — one rank, no accumulation, no trainable cond stage; the eager step runs otherwise."""
        if self.loss_type != 'l1':
            raise NotImplementedError("the training step is built for loss_type 'l1' (conf/diffusion/ldm_based.yaml)")
        tr = self._trainer_or_default()
        if t is None:
            t = torch.randint(0, self.num_timesteps, (x_start.shape[0],), device=x_start.device).long()
        noise = torch.randn_like(x_start) if noise is None else noise
        if cond_input is None:
            cond_input = self.__dict__.get("_last_cond_input")
        if tr.extra_params and cond_input is None:
            raise ValueError("the cond stage is in the optimizer (cond_stage_trainable): pass cond_input (the raw layout given to the cond stage) so "
                             "that its gradient can be computed")
        x_noisy = self.q_sample(x_start, t, noise)
        xc, ctx = self._split_cond(cond)
        after = None
        if cond_input is not None and hasattr(self.cond_stage_model, "backward"):
            nx = x_noisy.shape[1]
            after = lambda dx, dctx: self.cond_stage_model.backward(cond_input, dx[:, nx:].contiguous())
        if graph and after is None and not tr.extra_params and tr.accumulate_grad_batches == 1 and x_noisy.is_cuda:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
                return tr.train_step_graphed(x_noisy, xc, t, ctx, noise)
        return tr.train_step(x_noisy, xc, t, ctx, noise, group=group, after_backward=after)

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
        if isinstance(ckpt.get("optimizer_states"), (list, tuple)) and ckpt["optimizer_states"]:
            self.__dict__["_opt_loaded"] = ckpt["optimizer_states"][0]       # Lightning keeps torch's AdamW.state_dict() there
        self.restarted_from_ckpt = True
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
        self.__dict__["_ema_loaded"] = {name: ema[self._ema_name(name)] for name in named if self._ema_name(name) in ema}
        self.__dict__["_ema_num_updates"] = int(ema["num_updates"]) if "num_updates" in ema else 0
        tr = self.__dict__.get("_trainer")
        if tr is not None:      # a trainer built before the checkpoint arrived; configure_trainer() hands both over otherwise
            if self._ema_loaded:
                tr.load_ema({n[len("diffusion_model."):]: v for n, v in self._ema_loaded.items()}, self._ema_num_updates)
            if self.__dict__.get("_opt_loaded") is not None:
                tr.load_optimizer_state_dict(self._opt_loaded)
        return missing, unexpected

    def reference_state_dict(self, prefix: str = "_model.") -> dict:
        """State dict in the reference's key layout (see load_reference_state_dict), including `model_ema.*` when a trainer holds
        EMA shadows."""
        out = {prefix + k: v for k, v in self.state_dict().items()}
        tr = self.__dict__.get("_trainer")
        if tr is not None and tr.ema_named() is not None:
            out[prefix + "model_ema.decay"] = torch.tensor(tr.ema_decay, dtype=torch.float32)
            out[prefix + "model_ema.num_updates"] = torch.tensor(tr.ema_updates, dtype=torch.int)
            for name, v in tr.ema_named().items():
                out[prefix + "model_ema." + self._ema_name("diffusion_model." + name)] = v
        return out

    def reference_checkpoint(self, prefix: str = "_model.") -> dict:
        """{"state_dict", "optimizer_states"} as Lightning's ModelCheckpoint writes them for the reference (train_diff.py:64-66):
        parameters + `model_ema.*` in the reference's key layout, AdamW state in torch.optim.AdamW.state_dict() layout. Feeding it
        back through load_reference_state_dict + configure_trainer continues the run (EMA history, moments, bias-correction step)."""
        ck = {"state_dict": self.reference_state_dict(prefix)}
        tr = self.__dict__.get("_trainer")
        if tr is not None and tr._opt is not None:
            ck["optimizer_states"] = [tr.optimizer_state_dict()]
        return ck

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
            raise NotImplementedError("decode_first_stage: this LatentDiffusion was built without a first stage; pass first_stage_config "
                                      "(built as stedm_amd.vq.VQModelInterface) or a first_stage module")
        return self.first_stage_model.decode(z / self.scale_factor)


class S_ZSS_DM(LatentDiffusion):
    """networks/s_zss_dm.py:11-60 — LatentDiffusion + style aggregation block; `get_input` emits the hybrid conditioning
    dict {"c_concat": [layout], "c_crossattn": [style]}.

    `encoder` names the torchvision embedder of the mean/max/linear aggregators in the reference ("swin_v2_t", third-party,
    SURVEY.md §8c): the Swin-V2 family is built on the HIP kernels (stedm_amd/swin.py); any other embedder can be supplied as a module
    through `embedder=`. `style_agg: svit` and `style_sampling: none` need no embedder."""

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
                # s_zss_dm.py:19-20: `torchvision.models.get_model(encoder)` with its head replaced by Linear(768, 512). The HIP-backed
                # Swin-V2 (stedm_amd/swin.py) keeps torchvision's state-dict names, so a torchvision checkpoint loads into it; its arithmetic
                # is torchvision's published algorithm (third-party: parity unpinned). Other encoders: pass `embedder=`.
                from .swin import get_model
                embedder = get_model(encoder)
                embedder.head = torch.nn.Linear(768, 512)
            cls = {"linear": st.Agg_Linear, "max": st.Agg_Max, "mean": st.Agg_Mean}.get(name(agg_cfg))
            if cls is None:
                raise Exception("Unkown aggregation function!")
            self._agg_block = cls(sampling_cfg, embedder)
        self.register_module("agg_block", self._agg_block)

    def get_input(self, batch, k, cond_key=None, bs=None, predict_only: Optional[bool] = None, **kwargs):
        """s_zss_dm.py:45-60 over LatentDiffusion.get_input (ddpm.py:656-706). batch tensors are NHWC (LDM_Diffusion.prepare_batch,
        ldm_diffusion.py:51-60). Returns [z, {"c_concat": [c], "c_crossattn": [style]}].

        predict_only=True: z is a zero placeholder of the latent shape and the VQ encode is skipped — predict_step uses len(z) alone
        (ldm_diffusion.py:79-90), so the reference's two VQ encodes there are wasted work. Default (None): encode when a first stage
        exists (the reference's literal behaviour); without one, the placeholder in eval mode and an error in training mode (the
        latents are the training target: never silently zeros)."""
        skip_encode = predict_only is True or (predict_only is None and self.first_stage_model is None and not self.training)
        dev = self.device
        if skip_encode or self.first_stage_model is None:
            if not skip_encode:
                raise StedmHipError("S_ZSS_DM.get_input in training mode needs the first stage (the latents are the training target): pass "
                                    "first_stage_config; the zero placeholder exists for predict_step only")
            x = batch[k] if bs is None else batch[k][:bs]
            z = torch.zeros((x.shape[0], self.channels, self.image_size, self.image_size), device=dev)
            xc = batch[cond_key or self.cond_stage_key]
            xc = (xc if bs is None else xc[:bs]).permute(0, 3, 1, 2).float().contiguous().to(dev)
            outputs = [z, xc]
        else:
            self.cond_stage_trainable = True
            outputs = LatentDiffusion.get_input(self, batch, k, cond_key=cond_key, bs=bs, **kwargs)
            self.cond_stage_trainable = False
        self.cond_stage_trainable = False
        z, c = outputs[0], outputs[1]
        self.__dict__["_last_cond_input"] = c                # raw layout: the channel mapper's gradient needs it (cond_stage_trainable)
        with torch.no_grad():
            c = self.get_learned_conditioning(c)
            style_imgs = batch[self.embed_key]
            if bs is not None:
                style_imgs = style_imgs[:bs]
            style_features = self._agg_block(style_imgs.to(dev))
        noutputs = [z, {"c_concat": [c], "c_crossattn": [style_features]}]
        noutputs.extend(outputs[2:])
        return noutputs


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
                    style_sampling: str = "nearby", x_T: Optional[torch.Tensor] = None, dedup_uncond: bool = True, noises=None):
    """Lightning-free restatement of LDM_Diffusion.predict_step (modules/ldm_diffusion.py:76-91) up to the sampled latents:
    conditional get_input, unconditional batch {image: 0, segmentation: same, style_imgs: -2}, DDIM + CFG.

    dedup_uncond: the unconditional style input is the same constant (-2) image stack for every sample (ldm_diffusion.py:86) and the
    style encoder works per sample, so its output is one vector: it is computed for ONE sample and broadcast instead of running the
    encoder over the whole constant batch; the layout conditioning of the unconditional batch is the conditional one (same
    segmentation). False: the reference's literal second get_input."""
    z, c_0 = model.get_input(ldm_batch, "image", predict_only=True)
    kw = {} if x_T is None else {"x_T": x_T}
    if noises is not None:          # one N(0,1) tensor per DDIM iteration in place of the global-RNG draw of ddim.py:206 (eta > 0)
        kw["noises"] = noises
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
            z, c_uncond = model.get_input(unc_batch, "image", predict_only=True)
        out, _ = model.sample_log(c_0, batch_size=len(z), ddim=True, ddim_steps=ddim_steps, eta=eta, log_every_t=1000,
                                  unconditional_conditioning=c_uncond, unconditional_guidance_scale=cfg_scale, **kw)
    return out


@torch.no_grad()
def predict_latents_sharded(model: S_ZSS_DM, shard_batch: dict, global_batch: int, ddim_steps: int, eta: float = 0.0, cfg_scale: float = 1.0,
                            seed: int = 0, rank: Optional[int] = None, world: Optional[int] = None, group=None, gather: bool = True, **kw):
    """predict_step on one rank of a data-parallel prediction run (predict_diff.py:86: Trainer.predict under DDP hands every rank its shard of
    the dataset; modules/ldm_diffusion.py:76-107). `shard_batch` holds this rank's samples — the contiguous slice
    parallel.shard_range(global_batch, rank, world) of the global batch. Latents are independent, so nothing is exchanged inside the loop;
    the initial noise x_T (ddim.py:122) and, for eta > 0, every step's noise (ddim.py:206) come from per-SAMPLE streams keyed by the global
    sample id (parallel.per_sample_normal), so sample i is the same for every world size, which the reference's batch-shaped global-RNG
    draw cannot give. gather: all-gather the shards (RCCL over xGMI with backend "nccl") -> [global_batch, C, H, W] on every rank."""
    import torch.distributed as dist
    from . import parallel as par
    if rank is None or world is None:
        on = dist.is_available() and dist.is_initialized()
        rank, world = (dist.get_rank(group), dist.get_world_size(group)) if on else (0, 1)
    lo, hi = par.shard_range(int(global_batch), rank, world)
    ids = list(range(lo, hi))
    n = len(shard_batch["image"])
    if n != hi - lo:
        raise ValueError(f"rank {rank} of {world}: the shard holds {n} samples, shard_range({global_batch}) gives {hi - lo}")
    shape = (model.channels, model.image_size, model.image_size)
    dev = model.device
    on_gpu = torch.device(dev).type == "cuda"
    # (GPU: drawn on the device from the per-sample Philox key - one launch per tensor; the numpy streams remain for CPU tensors)
    draw = (lambda st: par.per_sample_normal_device(seed, lo, hi - lo, shape, st, dev)) if on_gpu else \
           (lambda st: par.per_sample_normal(seed, ids, shape, stream=st).to(dev))
    x_T = draw(0)
    noises = None
    if eta != 0.0:
        from .schedule import make_ddim_timesteps
        n_iter = int(make_ddim_timesteps(int(ddim_steps), model.num_timesteps).shape[0])      # (S = 6 -> 7 iterations: ddim.py's uniform stride)
        noises = [draw(1 + i) for i in range(n_iter)]
    from ._lib import StedmHipError
    err: Optional[StedmHipError] = None
    lat = None
    try:
        lat = predict_latents(model, shard_batch, ddim_steps, eta=eta, cfg_scale=cfg_scale, x_T=x_T, noises=noises, **kw)
    except StedmHipError as e:       # (the fp16 range guard raises at the end of the loop: ddim.py's f16_guard_check)
        if not (gather and world > 1):
            raise
        err = e
    if gather and world > 1:
        # a rank that raised must not leave the others waiting in the all-gather: agree on the outcome first (one 4-byte all-reduce)
        flag = torch.tensor([1 if err is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        if int(flag.item()):
            if err is not None:
                raise err
            raise StedmHipError(f"rank {rank} of {world}: another rank's sampling loop raised (fp16 operand overflow there); no latents gathered")
        lat = par.all_gather_samples(lat, int(global_batch), group)
    return lat


@torch.no_grad()
def images_for_saving(decoded: torch.Tensor, segmentation_nhwc: Optional[torch.Tensor] = None):
    """The array work of predict_step after decode_first_stage (modules/ldm_diffusion.py:93-99): decoded [B,3,H,W] fp32 -> uint8
    [B,H,W,3] (clip to [-1,1], scale, truncate), segmentation [B,H,W,ncls] -> uint8 class map. Returned on the device; PNG
    encoding stays with the caller."""
    img = ops.image_to_uint8(decoded.float().contiguous())
    seg = None if segmentation_nhwc is None else ops.argmax_u8(segmentation_nhwc.float().contiguous())
    return img, seg
