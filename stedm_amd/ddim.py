"""DDIM sampler with sequential-equivalent classifier-free guidance and std-rescale, HIP-backed.

Mirrors `ldm.models.diffusion.ddim.DDIMSampler` (reference ddim.py:11-210): same constructor,
`make_schedule`, `sample`, `ddim_sampling`, `p_sample_ddim` names / argument meaning / returns.
Host logic (schedule tables, the python loop) stays on the host as in the reference; every tensor
operation of the loop body runs in HIP kernels:
  * the two `apply_model` calls of ddim.py:177-178 -> one shared-encoder CFG pass (`apply_model_cfg`)
    when the model offers it, else two calls in the reference's order (cond, then uncond);
  * ddim.py:179-184 (CFG combine + (C,H)-std rescale, phi = 0.7) and :195-210 (x0 / direction / noise)
    -> one fused kernel (stedm_ddim_step);
  * with `use_graph=True` the whole step (timestep fill, U-Net, update, counter decrement) is captured
    once in a hipGraph and replayed per step, per-step scalars coming from a device table.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import ops
from .schedule import DDIMTables, make_ddim_tables


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        super().__init__()
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self.use_graph = bool(kwargs.get("use_graph", False))
        self._graph_cache = {}

    def register_buffer(self, name, attr):
        """ddim.py:18-22 moves tensors to "cuda"; here: to the model's device."""
        if isinstance(attr, torch.Tensor) and attr.device != self.model.device:
            attr = attr.to(self.model.device)
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        """ddim.py:24-53 — tables built on the host from the model's fp32 alphas_cumprod buffer."""
        acp = self.model.alphas_cumprod
        assert acp.shape[0] == self.ddpm_num_timesteps, 'alphas have to be defined for each timestep'
        tb: DDIMTables = make_ddim_tables(acp.detach().cpu().numpy(), ddim_num_steps, float(ddim_eta))
        self.tables = tb
        self.ddim_timesteps = tb.timesteps
        dev = self.model.device
        self.register_buffer('ddim_sigmas', torch.from_numpy(tb.sigmas))
        self.register_buffer('ddim_alphas', torch.from_numpy(tb.alphas))
        self.register_buffer('ddim_alphas_prev', torch.from_numpy(tb.alphas_prev))
        self.register_buffer('ddim_sqrt_one_minus_alphas', torch.from_numpy(tb.sqrt_one_minus_alphas))
        self._coefs = torch.from_numpy(tb.coef_table()).to(dev).contiguous()       # [n][4] device table
        self._ts_table = torch.from_numpy(tb.timesteps.astype(np.int64)).to(dev)    # [n] int64
        self._idx_all = torch.arange(tb.timesteps.shape[0], dtype=torch.int32, device=dev)
        self._eta = float(ddim_eta)

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.,
               unconditional_conditioning=None, **kwargs):
        """ddim.py:56-110."""
        if quantize_x0 or mask is not None or x0 is not None or score_corrector is not None or noise_dropout > 0. \
                or temperature != 1.:
            raise NotImplementedError("quantize_x0 / mask / x0 / score_corrector / noise_dropout / temperature: "
                                      "unused by the reference drivers (ldm_diffusion.py:82,90), not implemented")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        size = (batch_size, C, H, W)
        return self.ddim_sampling(conditioning, size, callback=callback, img_callback=img_callback, x_T=x_T,
                                  log_every_t=log_every_t, unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, noises=kwargs.get("noises"))

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, callback=None, img_callback=None, log_every_t=100,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, noises=None, **kwargs):
        """ddim.py:113-162. `noises` (optional list, one N(0,1) tensor per iteration) replaces the global-RNG
        draw of ddim.py:206 so that runs are reproducible across devices and shard counts."""
        device = self.model.device
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device).float().clone()
        timesteps = self.ddim_timesteps
        total_steps = timesteps.shape[0]
        intermediates = {'x_inter': [img.clone()], 'pred_x0': [img.clone()]}
        cfg = not (unconditional_conditioning is None or unconditional_guidance_scale == 1.)
        need_inter = lambda index: index % log_every_t == 0 or index == total_steps - 1

        if self.use_graph and callback is None and img_callback is None and self._eta == 0.0 \
                and hasattr(self.model, "apply_model_cfg"):
            out = self._sample_graph(img, cond, unconditional_conditioning, unconditional_guidance_scale, cfg,
                                     total_steps, log_every_t, intermediates)
            ops.f16_guard_check("the DDIM sampling loop")       # fp16 modes: raise rather than return samples computed through an inf
            return out

        pred_x0 = torch.empty_like(img)
        for i, step in enumerate(np.flip(timesteps)):
            index = total_steps - i - 1
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            nz = None
            if noises is not None:
                nz = noises[i].to(device).float().contiguous()
            elif self._eta != 0.0:
                nz = torch.randn(shape, device=device)     # ddim.py:206 (when sigma == 0 the draw cannot change x)
            img, pred_x0 = self.p_sample_ddim(img, cond, ts, index=index, unconditional_guidance_scale=unconditional_guidance_scale,
                                              unconditional_conditioning=unconditional_conditioning, _noise=nz, _out=(img, pred_x0),
                                              _uniform_t=True)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if need_inter(index):
                intermediates['x_inter'].append(img.clone())
                intermediates['pred_x0'].append(pred_x0.clone())
        ops.f16_guard_check("the DDIM sampling loop")
        return img, intermediates

    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, rescale_phi=0.7,
                      _noise: Optional[torch.Tensor] = None, _out=None, _uniform_t: bool = False):
        """ddim.py:164-210. Returns (x_prev, pred_x0)."""
        if use_original_steps or quantize_denoised or score_corrector is not None or repeat_noise:
            raise NotImplementedError("use_original_steps / quantize_denoised / score_corrector / repeat_noise not implemented")
        x = x.float().contiguous()
        e_u = None
        ours = hasattr(self.model, "apply_model_cfg")
        kw = {"uniform_t": True} if (ours and _uniform_t) else {}
        if unconditional_conditioning is None or unconditional_guidance_scale == 1.:
            e_c = self.model.apply_model(x, t, c, **kw)
        elif ours:
            e_c, e_u = self.model.apply_model_cfg(x, t, c, unconditional_conditioning, **kw)
        else:
            e_c = self.model.apply_model(x, t, c)                           # cond first, then uncond (ddim.py:177-178)
            e_u = self.model.apply_model(x, t, unconditional_conditioning)
        x_prev, pred_x0 = _out if _out is not None else (torch.empty_like(x), torch.empty_like(x))
        step = self._idx_all[index:index + 1]   # device-resident loop index (no H2D copy per step)
        ops.ddim_step(x, e_c.contiguous(), None if e_u is None else e_u.contiguous(), self._coefs, x_prev, pred_x0=pred_x0,
                      noise=_noise, step_idx=step, cfg_scale=float(unconditional_guidance_scale), rescale_phi=float(rescale_phi))
        return x_prev, pred_x0

    # ------------------------------------------------------------------------------------------------ graph replay
    def _sample_graph(self, img, cond, uncond, scale, cfg, total_steps, log_every_t, intermediates):
        """Only taken for eta == 0 (sigma == 0: the noise term of ddim.py:206 is identically zero)."""
        sg = StepGraph(self, img, cond, uncond if cfg else None, scale)

        def log(index):
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates['x_inter'].append(img.clone())
                intermediates['pred_x0'].append(sg.pred_x0.clone())

        sg.reset(total_steps - 1)
        sg.step_eager()   # packs weights and allocates every buffer before capture
        log(total_steps - 1)
        if total_steps > 1:
            with sg.stream_ctx():
                sg.capture()
                for i in range(1, total_steps):
                    sg.replay()
                    log(total_steps - i - 1)
            sg.join()
        return img, intermediates


class StepGraph:
    """One denoising step = {t fill from the device table, U-Net (shared-encoder CFG pass), fused DDIM/CFG update in
    place on `img`, device index decrement}, capturable once in a hipGraph and replayed for every step."""

    def __init__(self, sampler: DDIMSampler, img: torch.Tensor, cond, uncond, scale: float, rescale_phi: float = 0.7):
        self.s = sampler
        self.img = img
        self.cond, self.uncond, self.scale, self.phi = cond, uncond, float(scale), float(rescale_phi)
        dev = img.device
        b = img.shape[0]
        self.cfg = uncond is not None and scale != 1.0
        self.step = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.t_buf = torch.empty((b,), dtype=torch.int64, device=dev)
        self.pred_x0 = torch.empty_like(img)
        self.eps = torch.empty((2 * b if self.cfg else b,) + tuple(img.shape[1:]), dtype=torch.float32, device=dev)
        self.graph = None
        self.side = None

    def reset(self, index: int):
        self.step.fill_(int(index))

    def step_eager(self):
        s, m = self.s, self.s.model
        ops.step_set_t(s._ts_table, self.step, self.t_buf)
        if self.cfg:
            e_c, e_u = m.apply_model_cfg(self.img, self.t_buf, self.cond, self.uncond, out=self.eps, uniform_t=True)
        else:
            e_c, e_u = m.apply_model(self.img, self.t_buf, self.cond, out=self.eps, uniform_t=True), None
        ops.ddim_step(self.img, e_c, e_u, s._coefs, self.img, pred_x0=self.pred_x0, step_idx=self.step,
                      cfg_scale=self.scale, rescale_phi=self.phi)
        ops.step_advance(self.step, -1)

    def stream_ctx(self):
        if self.side is None:
            self.side = torch.cuda.Stream()
        self.side.wait_stream(torch.cuda.current_stream())
        return torch.cuda.stream(self.side)

    def join(self):
        torch.cuda.current_stream().wait_stream(self.side)

    def capture(self):
        """Must be called inside stream_ctx() after at least one step_eager()."""
        g = ops.Graph()
        with g:
            self.step_eager()
        self.graph = g

    def replay(self):
        self.graph.launch()
