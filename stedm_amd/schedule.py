"""Host-side (numpy) noise schedule and DDIM tables — the reference also builds these on the host
once per `sample()` call (ddim.py:24-53; util.py:21-74; ddpm.py:120-172). Written from the formulas,
with the reference's precision staging reproduced explicitly:

  betas            f64: linspace(sqrt(l0), sqrt(l1), T)**2                    (util.py:23-25)
  alphas_cumprod   f64 cumprod, then narrowed to the fp32 buffer               (ddpm.py:127-141)
  ddim tables      computed FROM the fp32 buffer (ddim.py:27-33,43-49): a_t = acp32[ts];
                   a_prev = [acp32[0]] + acp32[ts[:-1]];  (1 - a_t), 1/(1 - a_t) and sqrt(1 - a_t) evaluated
                   in fp32, the rest of the sigma formula in f64; all four per-step scalars reach the
                   device as fp32 (torch.full, ddim.py:195-198).
Bit-checked against the reference's own outputs in tests/test_host_logic.py (golden F2).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


def make_beta_schedule(n_timestep: int = 1000, linear_start: float = 1e-4, linear_end: float = 2e-2,
                       schedule: str = "linear") -> np.ndarray:
    if schedule != "linear":
        raise NotImplementedError(f"beta schedule {schedule!r}: only 'linear' is used by the reference configs")
    import torch  # host-side f64 linspace, the very call of util.py:23-25 (bit-exactness of the table matters)
    lin = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64).numpy()
    return lin ** 2


@dataclass
class NoiseSchedule:
    """The fp32 buffers of DDPM.register_schedule (ddpm.py:120-172) that the hot path reads."""
    num_timesteps: int
    betas: np.ndarray                        # f32
    alphas_cumprod: np.ndarray               # f32
    alphas_cumprod_prev: np.ndarray          # f32
    sqrt_alphas_cumprod: np.ndarray          # f32
    sqrt_one_minus_alphas_cumprod: np.ndarray  # f32

    @staticmethod
    def make(timesteps: int = 1000, linear_start: float = 1e-4, linear_end: float = 2e-2,
             beta_schedule: str = "linear") -> "NoiseSchedule":
        betas = make_beta_schedule(timesteps, linear_start, linear_end, beta_schedule)
        ac = np.cumprod(1.0 - betas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        f32 = lambda a: np.asarray(a, dtype=np.float64).astype(np.float32)
        return NoiseSchedule(int(timesteps), f32(betas), f32(ac), f32(ac_prev), f32(np.sqrt(ac)), f32(np.sqrt(1.0 - ac)))


def make_ddim_timesteps(num_ddim_timesteps: int, num_ddpm_timesteps: int = 1000, method: str = "uniform") -> np.ndarray:
    """util.py:46-60 — integer arithmetic, must be exact: c = T // S; ts = arange(0, T, c) + 1."""
    if method != "uniform":
        raise NotImplementedError("only the 'uniform' DDIM discretisation is used by the reference")
    c = num_ddpm_timesteps // num_ddim_timesteps
    return np.arange(0, num_ddpm_timesteps, c, dtype=np.int64) + 1


@dataclass
class DDIMTables:
    timesteps: np.ndarray   # int64 [n], ascending (ddim_timesteps)
    alphas: np.ndarray      # f32 [n]
    alphas_prev: np.ndarray  # f32 [n]
    sigmas: np.ndarray      # f32 [n]
    sqrt_one_minus_alphas: np.ndarray  # f32 [n]

    def coef_table(self) -> np.ndarray:
        """[n][4] fp32 rows {a_t, a_prev, sigma_t, sqrt(1 - a_t)} — the device table of stedm_ddim_step."""
        return np.stack([self.alphas, self.alphas_prev, self.sigmas, self.sqrt_one_minus_alphas], axis=1).astype(np.float32)


def make_ddim_tables(alphas_cumprod_f32: np.ndarray, S: int, eta: float = 0.0) -> DDIMTables:
    ts = make_ddim_timesteps(S, alphas_cumprod_f32.shape[0])
    acp = np.asarray(alphas_cumprod_f32, dtype=np.float32)
    a32 = acp[ts]                                                    # f32
    ap64 = np.concatenate([acp[:1], acp[ts[:-1]]]).astype(np.float64)  # f32 values held in f64
    one_minus_a = (np.float32(1.0) - a32)                            # evaluated in fp32 (torch f32 tensor op)
    ratio = a32.astype(np.float64) / ap64                            # f32 tensor / f64 array -> f64
    # ndarray / Tensor dispatches to Tensor.__rtruediv__ = reciprocal(fp32) * other: the reciprocal is rounded in fp32
    recip32 = (np.float32(1.0) / one_minus_a).astype(np.float64)
    sig64 = eta * np.sqrt(recip32 * (1.0 - ap64) * (1.0 - ratio))
    sq32 = np.sqrt(one_minus_a).astype(np.float32)                   # np.sqrt on the f32 tensor (ddim.py:49)
    return DDIMTables(ts, a32, ap64.astype(np.float32), sig64.astype(np.float32), sq32)
