"""Compact, order-sensitive summary of a large tensor for golden fixtures:
global moments plus 1024 values at fixed strided flat indices."""
import numpy as np
import torch

N_SAMPLES = 1024


def sample_index(numel: int) -> np.ndarray:
    n = min(N_SAMPLES, numel)
    return np.unique(np.linspace(0, numel - 1, n).astype(np.int64))


def summarize(t: torch.Tensor) -> dict:
    a = t.detach().cpu().double().reshape(-1).numpy()
    idx = sample_index(a.size)
    return {
        "shape": np.asarray(t.shape, dtype=np.int64),
        "mean": np.float64(a.mean()),
        "std": np.float64(a.std()),
        "abssum": np.float64(np.abs(a).sum()),
        "samples": a[idx].astype(np.float32),
    }


def check_summary(t: torch.Tensor, fx, key: str, rtol: float, what: str = ""):
    """Assert tensor `t` matches the summary stored under `key.*` in npz `fx` (relative to the tensor's std)."""
    s = summarize(t)
    assert tuple(s["shape"]) == tuple(fx[f"{key}.shape"]), (what, key, s["shape"], fx[f"{key}.shape"])
    scale = float(fx[f"{key}.std"]) + 1e-12
    err = np.abs(s["samples"].astype(np.float64) - fx[f"{key}.samples"].astype(np.float64)).max() / scale
    assert err <= rtol, f"{what} {key}: max sample err / std = {err:.3e} > {rtol}"
    assert abs(s["mean"] - float(fx[f"{key}.mean"])) <= rtol * scale, (what, key, "mean")
    assert abs(s["std"] - float(fx[f"{key}.std"])) <= rtol * scale, (what, key, "std")
    assert abs(s["abssum"] - float(fx[f"{key}.abssum"])) <= rtol * float(fx[f"{key}.abssum"]) + 1e-12, (what, key, "abssum")
    return err
