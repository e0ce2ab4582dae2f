"""Helpers shared by the parity tests: seeded sampling of big tensors against the committed fixtures."""
import os

import numpy as np
import torch

from transception_amd.seeded_init import _stream

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NSAMP = 2048


def sample_idx(tag: str, numel: int, n: int = NSAMP) -> np.ndarray:
    g = _stream(f"sample:{tag}", 3)
    return g.integers(0, numel, size=min(n, numel), dtype=np.int64)


def load(name: str):
    return np.load(os.path.join(GOLDEN, name))


def check_packed(store, tag: str, t: torch.Tensor, atol: float, rtol: float = 0.0, sum_rtol: float = 1e-4,
                 scale_rel: float = 0.0):
    """Compare tensor `t` with the (shape, samples, checksums) triple stored under `tag`."""
    a = t.detach().contiguous().float().cpu().reshape(-1).numpy()
    shape = tuple(int(s) for s in store[tag + "/shape"])
    assert tuple(t.shape) == shape, f"{tag}: shape {tuple(t.shape)} != golden {shape}"
    idx = sample_idx(tag, a.size)
    want = store[tag + "/samples"]
    got = a[idx]
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want) + scale_rel * float(np.abs(want).max())
    assert np.all(err <= tol), f"{tag}: max err {err.max():.3e} (tol {atol:g}+{rtol:g}*|x|), worst at {int(err.argmax())}"
    s, sa = store[tag + "/sum"]
    gsa = np.abs(a.astype(np.float64)).sum()
    assert abs(gsa - sa) <= sum_rtol * max(sa, 1e-12) + atol * a.size * 0.01, f"{tag}: |x| checksum {gsa} vs {sa}"
    return float(err.max())
