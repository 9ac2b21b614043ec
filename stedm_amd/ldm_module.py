"""In-tree counterpart of the reference's Lightning wrapper `modules/ldm_diffusion.py::LDM_Diffusion` (:14-234) for the HIP path.

Same constructor argument (the Hydra config), same members the drivers call — `prepare_batch` (:51-60), `training_step(batch,
batch_idx)` (:63-73), `predict_step` (:76-107), `on_train_batch_start/end` (:110-115), `configure_optimizers` (:224-234) — with one
declared difference: `automatic_optimization = False`. The training step runs the fused HIP path (`S_ZSS_DM.training_step_hip`:
forward + L1 + hand-scheduled backward, gradient accumulation, bucketed all-reduce over the ranks, fused AdamW + EMA), so Lightning's
own backward / DDP reducer / optimizer loop have nothing to do; `train_diff.py` can construct this class in place of the reference's
and call `Trainer.fit` with `strategy=DDPStrategy(...)` for the process group alone. Where pytorch_lightning is not installed (this
repository's test boxes) the class derives from `torch.nn.Module` and is driven by a plain loop that makes the same calls in the same
order (tests/test_gpu_train.py::test_ldm_module_training_loop_as_the_reference_drives_it).

A maintainer who wants to keep Lightning's automatic optimisation and torch.optim.AdamW instead uses `S_ZSS_DM.training_step` (the
reference's own seam, autograd bridge in latent_diffusion.py) and `attach_optimizer` for the cross-rank average.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .latent_diffusion import S_ZSS_DM, images_for_saving, predict_latents, prepare_batch

try:  # pragma: no cover - not installed on the test boxes
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except ImportError:
    pl = None
    _Base = nn.Module


class _Cfg(dict):
    """attribute access over a plain nested dict (the reference passes an OmegaConf DictConfig; both work)"""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return _Cfg(v) if isinstance(v, dict) and not isinstance(v, _Cfg) else v


def _to_container(c):
    if isinstance(c, dict):
        return {k: _to_container(v) for k, v in c.items()}
    try:  # OmegaConf, when present
        from omegaconf import OmegaConf
        return OmegaConf.to_container(c)
    except ImportError:
        return c


class LDM_Diffusion(_Base):
    def __init__(self, cfg, wandb_id: str = "", accumulate_grad_batches: int = 4):
        super().__init__()
        cfg = _Cfg(cfg) if isinstance(cfg, dict) and not isinstance(cfg, _Cfg) else cfg
        self._cfg = cfg
        self._lr = cfg.lr
        self._wandb_id = wandb_id
        self.automatic_optimization = False
        # Trainer(accumulate_grad_batches=4) is hard-coded in train_diff.py:76; with manual optimisation the module owns the window
        self._accumulate = int(accumulate_grad_batches)
        ldm_dict = dict(_to_container(cfg.diffusion))
        fs = ldm_dict.get("first_stage_config")
        if isinstance(fs, dict) and isinstance(fs.get("params"), dict) and fs["params"].get("ckpt_path") and hasattr(cfg, "location"):
            fs["params"]["ckpt_path"] = cfg.location.result_dir + "/" + fs["params"]["ckpt_path"]      # ldm_diffusion.py:30
        ldm_dict.pop("ckpt_path", None)
        self._model = S_ZSS_DM(encoder="swin_v2_t", sampling_cfg=cfg.style_sampling, agg_cfg=cfg.style_agg, cfg=cfg, **ldm_dict)
        self.register_module("model", self._model)          # state-dict aliasing of ldm_diffusion.py:38-41: `_model.*` and `model.*`
        self._loss_sum, self._loss_n = None, 0          # device-side accumulator (the reference's MeanMetric, ldm_diffusion.py:44)
        self.predict_dir: Optional[str] = None

    def forward(self, x, *args, **kwargs):
        return self._model.forward(x, *args, **kwargs)

    def prepare_batch(self, batch):
        """ldm_diffusion.py:51-60."""
        return prepare_batch(batch, device=self._model.device)

    # ------------------------------------------------------------------------------------------ training
    def configure_optimizers(self):
        """ldm_diffusion.py:224-234: AdamW(lr) over model.model (+ cond_stage_model while cond_stage_trainable; the aggregation block is
        not in the reference's list). Built as the fused HIP optimizer; Lightning gets no optimizer object (manual optimisation)."""
        self._model.configure_trainer(lr=self._lr, accumulate_grad_batches=self._accumulate)
        return None

    def training_step(self, batch, batch_idx):
        """ldm_diffusion.py:63-73 with the fused step in place of `loss = self._model.training_step(ldm_batch, batch_idx)` + Lightning's
        backward / optimizer loop."""
        ldm_batch = self.prepare_batch(batch)
        m = self._model
        if m.__dict__.get("_trainer") is None:
            self.configure_optimizers()
        x, c = m.get_input(ldm_batch, m.first_stage_key)[:2]
        loss = m.training_step_hip(x, c)
        # MeanMetric of ldm_diffusion.py:44,71: accumulated on the device — no host sync per micro-batch; train_loss() converts once
        ld = loss.detach().float().reshape(())
        self._loss_sum = ld.clone() if self._loss_sum is None else self._loss_sum + ld
        self._loss_n += 1
        return loss

    def on_train_batch_start(self, batch, batch_idx):
        """ldm_diffusion.py:110-112 -> ddpm.py:479-494, which returns at once unless scale_by_std is set and this is batch 0: the batch is
        only prepared (a seg_merge kernel plus copies) when that rescale will actually run."""
        m = self._model
        if not (getattr(m, "scale_by_std", False) and batch_idx == 0):
            return
        m.on_train_batch_start(self.prepare_batch(batch), batch_idx, -1)

    def on_train_batch_end(self, *args, **kwargs):
        self._model.on_train_batch_end(*args, **kwargs)

    def train_loss(self, reset: bool = True) -> float:
        """what on_train_epoch_end logs as "Train Loss" (ldm_diffusion.py:118-120)"""
        v = 0.0 if self._loss_sum is None else float(self._loss_sum) / max(1, self._loss_n)
        if reset:
            self._loss_sum, self._loss_n = None, 0
        return v

    # ------------------------------------------------------------------------------------------ prediction
    @torch.no_grad()
    def predict_step(self, batch, batch_idx):
        """ldm_diffusion.py:76-107: conditional + unconditional conditioning, DDIM + CFG, VQ decode, uint8 images and class maps;
        PNG files when `predict_dir` is set. Returns (images [B,H,W,3] uint8, segmentation [B,H,W] uint8) on the host."""
        cfg = self._cfg
        ldm_batch = self.prepare_batch(batch)
        sname = cfg.style_sampling["name"] if isinstance(cfg.style_sampling, dict) else cfg.style_sampling.name
        lat = predict_latents(self._model, ldm_batch, ddim_steps=cfg.ddim_steps, eta=cfg.eta, cfg_scale=cfg.cfg_scale, style_sampling=sname)
        dec = self._model.decode_first_stage(lat)
        img, seg = images_for_saving(dec, ldm_batch["segmentation"])
        img, seg = img.cpu().numpy(), seg.cpu().numpy()
        if self.predict_dir is not None and len(batch) > 4:
            from PIL import Image
            for im, sg, num in zip(img, seg, batch[4].cpu().numpy()):
                num_str = str(int(num)).zfill(5)
                Image.fromarray(im).save(os.path.join(self.predict_dir, f"img_{num_str}.png"))
                Image.fromarray(sg).save(os.path.join(self.predict_dir, f"seg_{num_str}.png"))
        return img, seg
