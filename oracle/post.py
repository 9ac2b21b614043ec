"""TEST INFRASTRUCTURE ONLY — numpy restatement of predict_step's array epilogue (modules/ldm_diffusion.py:93-99); integer
outputs, compared bit-exactly. Only tests/, smoke() and bench.py's cpu_baseline leg may import this package."""
import numpy as np


def image_to_uint8(x_nchw: np.ndarray) -> np.ndarray:
    """ldm_diffusion.py:93-95: torch.clip(x, -1, 1) -> permute(0,2,3,1) -> (+1) * 127.5 in float32 -> astype(uint8) (truncation)."""
    x = np.clip(x_nchw.astype(np.float32), np.float32(-1), np.float32(1))
    return ((np.transpose(x, (0, 2, 3, 1)) + 1) * 127.5).astype(np.uint8)


def segmentation_to_uint8(seg_nhwc: np.ndarray) -> np.ndarray:
    """ldm_diffusion.py:98: torch.argmax(seg, dim=-1) -> uint8 (first maximum on ties)."""
    return np.argmax(seg_nhwc, axis=-1).astype(np.uint8)
