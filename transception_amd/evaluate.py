"""Evaluation of the TransCeption path on MI355X: the device side of the reference's
`test_single_volume` (utils.py:63-110) and `inference` (test.py:60-86).

The reference runs one slice at a time (batch 1, host sync per slice, `utils.py:67-88`).  Here the slices of a volume are
normalised and pushed through the eval-mode forward in batches, `argmax(softmax(logits))` and the per-class voxel counts of
the Dice score are one HIP kernel (`tc_argmax_counts`), and only the uint8 label map returns to the host.

Host-side pieces that stay on the CPU exactly as in the reference: the order-3 `scipy.ndimage.zoom` of a slice to the network
size and the order-0 zoom of the prediction back (`utils.py:69-70,83-84`), and HD95 (medpy, not available here).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch

from ._lib import TC_F32, lib


def argmax_counts(logits: torch.Tensor, labels: Optional[torch.Tensor] = None,
                  counts: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """pred[b,h,w] = argmax_k logits[b,k,h,w] (uint8).  With `labels` (int64 [B,H,W]) the per-class counts
    (|pred==k & gt==k|, |pred==k|, |gt==k|) are ADDED to `counts` (float32 [ncls,3], created zeroed when None)."""
    if not logits.is_cuda:
        raise RuntimeError("transception_amd.evaluate runs on MI355X only (no CPU fallback)")
    B, C, H, W = logits.shape
    lg = logits.contiguous().float()
    pred = torch.empty((B, H, W), dtype=torch.uint8, device=lg.device)
    lab = None
    if labels is not None:
        lab = labels.contiguous().long()
        if counts is None:
            counts = torch.zeros((C, 3), dtype=torch.float32, device=lg.device)
    stream = torch.cuda.current_stream(lg.device).cuda_stream
    lib().tc_argmax_counts(lg.data_ptr(), lab.data_ptr() if lab is not None else None, pred.data_ptr(),
                           counts.data_ptr() if counts is not None else None, B, C, H * W, TC_F32, stream)
    return pred, counts


def dice_from_counts(counts: np.ndarray) -> List[float]:
    """Per-class Dice with calculate_metric_percase's conventions (utils.py:50-60) for classes 1..C-1:
    2|P&G|/(|P|+|G|) when both are non-empty, 1 when only the prediction is non-empty, else 0."""
    out = []
    for inter, p, g in np.asarray(counts, dtype=np.float64)[1:]:
        if p > 0 and g > 0:
            out.append(float(2.0 * inter / (p + g)))
        elif p > 0:
            out.append(1.0)
        else:
            out.append(0.0)
    return out


@torch.no_grad()
def predict_slices(model, slices: torch.Tensor, batch: int = 16) -> torch.Tensor:
    """slices: float [N,H,W] in [0,1] at the network size (a multiple of 32); returns uint8 [N,H,W] labels.
    Normalisation (x-0.5)/0.5 as `transforms.Normalize([0.5],[0.5])` (utils.py:71-75); eval-mode BatchNorm (utils.py:78)."""
    was_training = model.training
    model.eval()
    out = torch.empty(slices.shape, dtype=torch.uint8, device=slices.device)
    try:
        for i in range(0, slices.shape[0], batch):
            x = ((slices[i:i + batch].float() - 0.5) / 0.5).unsqueeze(1)
            out[i:i + batch] = argmax_counts(model(x))[0]
    finally:
        model.train(was_training)
    return out


@torch.no_grad()
def evaluate_volume(model, image: np.ndarray, label: np.ndarray, classes: int = 9, patch_size=(224, 224),
                    batch: int = 16) -> List[float]:
    """`test_single_volume` for one [D,H,W] volume (utils.py:63-98): per-class Dice for classes 1..classes-1.
    Slices are zoomed on the host exactly as the reference does; everything between runs on the GPU."""
    from scipy.ndimage import zoom
    dev = next(model.parameters()).device
    D, X, Y = image.shape
    resize = (X, Y) != tuple(patch_size)
    sl = np.stack([zoom(image[d], (patch_size[0] / X, patch_size[1] / Y), order=3) if resize else image[d] for d in range(D)])
    pred = predict_slices(model, torch.from_numpy(sl.astype(np.float32)).to(dev), batch).cpu().numpy()
    if resize:
        pred = np.stack([zoom(pred[d], (X / patch_size[0], Y / patch_size[1]), order=0) for d in range(D)])
    counts = np.zeros((classes, 3), dtype=np.float64)
    for k in range(classes):
        p, g = pred == k, label == k
        counts[k] = (np.logical_and(p, g).sum(), p.sum(), g.sum())
    return dice_from_counts(counts)
