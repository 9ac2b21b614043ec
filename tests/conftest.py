import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))

    return load


@pytest.fixture(scope="module", autouse=True)
def _poison_free_gpu_memory():
    """GPU tier: before every test module the caching allocator's free blocks are filled with NaN, so that a kernel reading memory it was
    meant to overwrite (an accumulator opened with beta = 0, a pad row nobody wrote) shows up as NaN instead of passing on whatever a freed
    tensor left there — how the accumulation-window bug of round 4 surfaced by accident. No-op without a GPU."""
    try:
        import torch
    except Exception:
        yield
        return
    if torch.cuda.is_available():
        big = [torch.full((256 << 20,), float("nan"), device="cuda") for _ in range(6)]          # 6 x 1 GiB for the large-block pool
        small = [torch.full((s,), float("nan"), device="cuda") for s in (1 << 10, 1 << 13, 1 << 16, 1 << 18) for _ in range(128)]
        del big, small
    yield
