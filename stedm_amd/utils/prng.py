"""Deterministic, torch-RNG-independent parameter / input recipe.

The reference ships no weights we can use (no network, no checkpoints) and every
``zero_module`` layer makes random-init outputs degenerate (SURVEY.md §8c trap 13), so
benchmarks, golden fixtures and parity tests all fill parameters from this recipe:
a numpy Philox stream keyed by (seed, crc32(parameter name)).  The same recipe is
applied to the reference modules (tests/golden/make_golden.py), to the CPU oracle and
to the HIP-backed modules, so all three see bit-identical fp32 parameters.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, zlib.crc32(name.encode())]))


def normal(seed: int, name: str, shape, std: float = 1.0, mean: float = 0.0) -> torch.Tensor:
    """fp32 N(mean, std) tensor of `shape`, keyed by (seed, name)."""
    a = _rng(seed, name).standard_normal(size=tuple(shape), dtype=np.float32)
    return torch.from_numpy(a * np.float32(std) + np.float32(mean))


def uniform(seed: int, name: str, shape, lo: float = -1.0, hi: float = 1.0) -> torch.Tensor:
    a = _rng(seed, name).random(size=tuple(shape), dtype=np.float32)
    return torch.from_numpy(a * np.float32(hi - lo) + np.float32(lo))


def _is_norm_name(name: str) -> bool:
    # GroupNorm / LayerNorm affine parameters in the reference state-dict layouts:
    #   U-Net:  *.in_layers.0.*, *.out_layers.0.*, out.0.*, *.norm.*   (openaimodel.py:214-245,325,729-733)
    #   SpatialTransformer: *.norm.*, *.norm1/2/3.*                    (attention.py:203-205,231)
    #   sViT: *.norm.* (PreNorm), to_patch_tokens.1.*, mlp_head.0.*    (vit_set.py:17,94,140)
    parts = name.split(".")
    if len(parts) < 2:
        return False
    leaf_parent = ".".join(parts[:-1])
    return (
        leaf_parent.endswith("in_layers.0")
        or leaf_parent.endswith("out_layers.0")
        or leaf_parent == "out.0"
        or parts[-2] in ("norm", "norm1", "norm2", "norm3", "norm_out")     # norm_out: first-stage Encoder / Decoder (model.py:420, 519)
        or leaf_parent.endswith("to_patch_tokens.1")
        or leaf_parent.endswith("mlp_head.0")
    )


def fill_value(seed: int, name: str, shape) -> torch.Tensor:
    """Value for one named parameter.

    * norm affine: weight = 1 + 0.1 N(0,1), bias = 0.1 N(0,1)
    * scalar `temperature` (LSA, vit_set.py:40): left to the caller (kept at its init)
    * weights with >= 2 dims: N(0, 1/sqrt(fan_in))  (fan_in = prod(shape[1:]))
    * biases / other 1-d: 0.05 N(0,1)
    * pos_embedding / cls_token (vit_set.py:130-131): 0.5 N(0,1)
    Every zero-initialised layer of the reference is thereby re-randomised.
    """
    shape = tuple(shape)
    leaf = name.split(".")[-1]
    if _is_norm_name(name):
        if leaf == "weight":
            return normal(seed, name, shape, std=0.1, mean=1.0)
        return normal(seed, name, shape, std=0.1)
    if leaf in ("pos_embedding", "cls_token"):
        return normal(seed, name, shape, std=0.5)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return normal(seed, name, shape, std=1.0 / np.sqrt(fan_in))
    return normal(seed, name, shape, std=0.05)


@torch.no_grad()
def fill_state_dict(shapes: dict, seed: int = 0, skip=("temperature",)) -> dict:
    """{name: shape} -> {name: fp32 tensor}. Names whose leaf is in `skip` are omitted."""
    out = {}
    for name, shape in shapes.items():
        if name.split(".")[-1] in skip:
            continue
        out[name] = fill_value(seed, name, shape)
    return out


@torch.no_grad()
def fill_module_(module: torch.nn.Module, seed: int = 0, skip=("temperature",)) -> None:
    """In-place fill of every parameter of `module` from the recipe (by state-dict name)."""
    for name, p in module.named_parameters():
        if name.split(".")[-1] in skip:
            continue
        p.copy_(fill_value(seed, name, p.shape).to(p.device, p.dtype))
