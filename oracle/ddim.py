"""ORACLE (test infrastructure, NOT product code): numpy/torch-CPU restatement of the
reference's noise schedule, DDIM sampler with sequential CFG + std-rescale, q_sample
and the L1 training loss.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Reference lines followed (all under /root/reference/):
  ldm/modules/diffusionmodules/util.py:21-43   make_beta_schedule ("linear")
  ldm/modules/diffusionmodules/util.py:46-60   make_ddim_timesteps ("uniform", +1 offset)
  ldm/modules/diffusionmodules/util.py:63-74   make_ddim_sampling_parameters
  ldm/models/diffusion/ddpm.py:120-172         DDPM.register_schedule (f64 numpy -> f32 buffers)
  ldm/models/diffusion/ddpm.py:277-280         q_sample
  ldm/models/diffusion/ddpm.py:282-295, 1030-1040  L1 loss, mean over (C,H,W) then batch
  ldm/models/diffusion/ddim.py:24-53           DDIMSampler.make_schedule
  ldm/models/diffusion/ddim.py:113-162         ddim_sampling loop
  ldm/models/diffusion/ddim.py:164-210         p_sample_ddim (CFG cond-then-uncond, rescale over dims (1,2))
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np
import torch


def make_beta_schedule(n_timestep: int = 1000, linear_start: float = 0.0015, linear_end: float = 0.0205) -> np.ndarray:
    """util.py:21-25 — linspace of sqrt(beta) in f64, squared. Defaults = conf/diffusion/ldm_based.yaml:1-3."""
    return (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2).numpy()


class Schedule:
    """ddpm.py:120-172 — the fp32 buffers the sampler and q_sample read."""

    def __init__(self, timesteps: int = 1000, linear_start: float = 0.0015, linear_end: float = 0.0205):
        betas = make_beta_schedule(timesteps, linear_start, linear_end)
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        f32 = lambda a: torch.tensor(a, dtype=torch.float32)
        self.num_timesteps = int(timesteps)
        self.betas = f32(betas)
        self.alphas_cumprod = f32(ac)
        self.alphas_cumprod_prev = f32(ac_prev)
        self.sqrt_alphas_cumprod = f32(np.sqrt(ac))
        self.sqrt_one_minus_alphas_cumprod = f32(np.sqrt(1.0 - ac))


def make_ddim_timesteps(num_ddim_timesteps: int, num_ddpm_timesteps: int = 1000) -> np.ndarray:
    """util.py:46-60 — c = T // S (so S=128 gives 143 steps), then +1."""
    c = num_ddpm_timesteps // num_ddim_timesteps
    return np.asarray(list(range(0, num_ddpm_timesteps, c))) + 1


def make_ddim_sampling_parameters(alphacums: torch.Tensor, ddim_timesteps: np.ndarray, eta: float):
    """util.py:63-74 — operates on the fp32 alphas_cumprod buffer (ddim.py:43 passes .cpu() of the f32 buffer)."""
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return sigmas, alphas, alphas_prev


class DDIMSchedule:
    """ddim.py:24-53 — per-run tables. alphas stays a torch f32 tensor, alphas_prev / sigmas become numpy
    f64 arrays exactly as in the reference (they are narrowed to fp32 by torch.full at ddim.py:195-198)."""

    def __init__(self, sched: Schedule, S: int, eta: float = 0.0):
        self.ddim_timesteps = make_ddim_timesteps(S, sched.num_timesteps)
        sig, a, ap = make_ddim_sampling_parameters(sched.alphas_cumprod, self.ddim_timesteps, eta)
        self.ddim_sigmas = sig                      # torch f32 tensor (eta * sqrt(tensor expr))
        self.ddim_alphas = a                        # torch f32 tensor
        self.ddim_alphas_prev = ap                  # numpy f64 array
        self.ddim_sqrt_one_minus_alphas = np.sqrt(1.0 - a)  # ddim.py:49 (np.sqrt on a torch tensor -> tensor)

    def scalars(self, index: int):
        """The four per-step scalars exactly as fp32 values (ddim.py:195-198)."""
        f = lambda v: float(torch.full((1,), float(v), dtype=torch.float32)[0])
        return (f(self.ddim_alphas[index]), f(self.ddim_alphas_prev[index]),
                f(self.ddim_sigmas[index]), f(self.ddim_sqrt_one_minus_alphas[index]))


@torch.no_grad()
def cfg_combine(e_t: torch.Tensor, e_t_uncond: torch.Tensor, scale: float, rescale_phi: float = 0.7) -> torch.Tensor:
    """ddim.py:179-184 — rescaled CFG; std is unbiased over dims (1,2) = (C,H), keepdim -> [B,1,1,W]."""
    e_w = e_t_uncond + scale * (e_t - e_t_uncond)
    dims = tuple(range(1, e_t.ndim - 1))
    rescaled = e_w * (e_t.std(dim=dims, keepdim=True) / e_w.std(dim=dims, keepdim=True))
    return rescaled * rescale_phi + (1.0 - rescale_phi) * e_t


@torch.no_grad()
def ddim_update(x: torch.Tensor, e_t: torch.Tensor, a_t: float, a_prev: float, sigma_t: float,
                sqrt_one_minus_at: float, noise: Optional[torch.Tensor] = None):
    """ddim.py:195-210 with temperature=1, no quantisation, no dropout. `noise` ~ N(0,1) (drawn every step
    in the reference even when sigma=0, ddim.py:206); None means sigma*noise contributes exactly 0."""
    b = x.shape[0]
    full = lambda v: torch.full((b, 1, 1, 1), v, dtype=torch.float32)
    A, AP, SG, SQ = full(a_t), full(a_prev), full(sigma_t), full(sqrt_one_minus_at)
    pred_x0 = (x - SQ * e_t) / A.sqrt()
    dir_xt = (1.0 - AP - SG ** 2).sqrt() * e_t
    nz = SG * (noise if noise is not None else torch.zeros_like(x))
    x_prev = AP.sqrt() * pred_x0 + dir_xt + nz
    return x_prev, pred_x0


@torch.no_grad()
def ddim_sample(apply_model: Callable, sched: Schedule, x_T: torch.Tensor, cond, S: int, eta: float = 0.0,
                uncond=None, scale: float = 1.0, noises: Optional[list] = None, rescale_phi: float = 0.7,
                trace: Optional[list] = None) -> torch.Tensor:
    """ddim.py:113-162 + 164-210. apply_model(x, t[int64 B], cond) -> eps. `noises[i]` is the N(0,1) draw of
    loop iteration i (injected, since RNG streams cannot be matched across devices)."""
    ds = DDIMSchedule(sched, S, eta)
    ts = ds.ddim_timesteps
    total = ts.shape[0]
    img = x_T
    b = x_T.shape[0]
    for i, step in enumerate(np.flip(ts)):
        index = total - i - 1
        t = torch.full((b,), int(step), dtype=torch.long)
        if uncond is None or scale == 1.0:
            e_t = apply_model(img, t, cond)
        else:
            e_c = apply_model(img, t, cond)        # cond first, then uncond (ddim.py:177-178)
            e_u = apply_model(img, t, uncond)
            e_t = cfg_combine(e_c, e_u, scale, rescale_phi)
        a_t, a_prev, sig, sq = ds.scalars(index)
        img, pred_x0 = ddim_update(img, e_t, a_t, a_prev, sig, sq, None if noises is None else noises[i])
        if trace is not None:
            trace.append((img.clone(), pred_x0.clone()))
    return img


@torch.no_grad()
def q_sample(sched: Schedule, x_start: torch.Tensor, t: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """ddpm.py:277-280 with extract_into_tensor (util.py:96-99)."""
    sh = (x_start.shape[0],) + (1,) * (x_start.ndim - 1)
    return (sched.sqrt_alphas_cumprod.gather(-1, t).reshape(sh) * x_start
            + sched.sqrt_one_minus_alphas_cumprod.gather(-1, t).reshape(sh) * noise)


def l1_loss(model_output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """ddpm.py:1030-1040 with loss_type l1, logvar == 0, l_simple_weight 1, original_elbo_weight 0."""
    return (target - model_output).abs().mean(dim=[1, 2, 3]).mean()
