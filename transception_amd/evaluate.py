"""Evaluation of the TransCeption path on MI355X: the device side of the reference's
`test_single_volume` (utils.py:63-110) and `inference` (test.py:60-86).

The reference runs one slice at a time (batch 1, host sync per slice, `utils.py:67-88`).  Here the slices of a volume are
normalised and pushed through the eval-mode forward in batches, `argmax(softmax(logits))` and the per-class voxel counts of
the Dice score are one HIP kernel (`tc_argmax_counts`), and only the uint8 label map returns to the host.

Host-side pieces that stay on the CPU exactly as in the reference: the order-3 `scipy.ndimage.zoom` of a slice to the network
size and the order-0 zoom of the prediction back (`utils.py:69-70,83-84`), and HD95: `medpy.metric.binary.hd95` (`utils.py:55`;
medpy is not installable here) restated from its published algorithm on scipy.ndimage -- surface voxels = mask minus its
erosion (connectivity 1), distances by the Euclidean distance transform of the other mask's surface, 95th percentile of both
directions pooled.  Parity with medpy itself is unpinned; tests pin it to a brute-force evaluation of that definition.
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


def surface_distances(result: np.ndarray, reference: np.ndarray, voxelspacing=None, connectivity: int = 1) -> np.ndarray:
    """Distances from every surface voxel of `result` to the nearest surface voxel of `reference` (medpy __surface_distances)."""
    from scipy.ndimage import binary_erosion, distance_transform_edt, generate_binary_structure
    result, reference = np.atleast_1d(result.astype(bool)), np.atleast_1d(reference.astype(bool))
    if not result.any() or not reference.any():
        raise RuntimeError("surface distances need non-empty masks")
    footprint = generate_binary_structure(result.ndim, connectivity)
    result_border = result ^ binary_erosion(result, structure=footprint, iterations=1)
    reference_border = reference ^ binary_erosion(reference, structure=footprint, iterations=1)
    dt = distance_transform_edt(~reference_border, sampling=voxelspacing)
    return dt[result_border]


def hd95(result: np.ndarray, reference: np.ndarray, voxelspacing=None, connectivity: int = 1) -> float:
    """95th percentile of the symmetric surface distances (medpy.metric.binary.hd95, called at utils.py:55)."""
    a = surface_distances(result, reference, voxelspacing, connectivity)
    b = surface_distances(reference, result, voxelspacing, connectivity)
    return float(np.percentile(np.hstack((a, b)), 95))


def calculate_metric_percase(pred: np.ndarray, gt: np.ndarray) -> Tuple[float, float]:
    """(dice, hd95) of one class of one volume with the reference's conventions (utils.py:50-60)."""
    pred, gt = pred > 0, gt > 0
    ps, gs = int(pred.sum()), int(gt.sum())
    if ps > 0 and gs > 0:
        return float(2.0 * np.logical_and(pred, gt).sum() / (ps + gs)), hd95(pred, gt)
    if ps > 0:
        return 1.0, 0.0
    return 0.0, 0.0


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
def zoom_volume_to_network(vol: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """[D,X,Y] float32 slices on the GPU -> [D,size] : scipy.ndimage.zoom(slice, (size/X, size/Y), order=3) of utils.py:69-70 as the
    device spline prefilter + 4x4-tap evaluation of the input pipeline (csrc/data.hip; within 1e-6 of scipy, tests/test_data_gpu.py)."""
    L = lib()
    D, X, Y = vol.shape
    stream = torch.cuda.current_stream(vol.device).cuda_stream
    vol = vol.contiguous()
    coef = torch.empty((D, X, Y), dtype=torch.float64, device=vol.device)
    out = torch.empty((D, 1, size[0], size[1]), dtype=torch.float32, device=vol.device)
    L.tc_spline_prefilter(vol.data_ptr(), coef.data_ptr(), D, X, Y, stream)
    L.tc_zoom_normalize(coef.data_ptr(), vol.data_ptr(), None, out.data_ptr(), None, D, X, Y, size[0], size[1], 0.0, 1.0, stream)
    return out[:, 0]


@torch.no_grad()
def zoom_labels(pred: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """uint8 [D,h,w] label maps -> [D,size]: scipy.ndimage.zoom(pred, ..., order=0) of utils.py:83-84 (nearest sample at
    o (in-1)/(out-1); bit-exact integer work) through the label path of the same device kernel."""
    L = lib()
    D, h, w = pred.shape
    dev = pred.device
    stream = torch.cuda.current_stream(dev).cuda_stream
    pred = pred.contiguous()
    coef = torch.zeros(D * h * w, dtype=torch.float64, device=dev)            # the image half of the kernel is unused
    x = torch.empty((D, 1, size[0], size[1]), dtype=torch.float32, device=dev)
    y = torch.empty((D, size[0], size[1]), dtype=torch.int64, device=dev)
    L.tc_zoom_normalize(coef.data_ptr(), None, pred.data_ptr(), x.data_ptr(), y.data_ptr(), D, h, w, size[0], size[1], 0.0, 1.0, stream)
    return y.to(torch.uint8)


@torch.no_grad()
def evaluate_volume(model, image: np.ndarray, label: np.ndarray, classes: int = 9, patch_size=(224, 224),
                    batch: int = 16, with_hd95: bool = False, host_zoom: bool = False):
    """`test_single_volume` for one [D,H,W] volume (utils.py:63-98): per-class Dice for classes 1..classes-1, or with
    `with_hd95` the reference's metric_list of (dice, hd95) pairs.  The volume goes to the GPU once; the order-3 zoom to the network
    size, inference, argmax and the order-0 zoom back all run there (host_zoom=True: scipy per slice, as the reference does)."""
    dev = next(model.parameters()).device
    D, X, Y = image.shape
    resize = (X, Y) != tuple(patch_size)
    if host_zoom:
        from scipy.ndimage import zoom
        sl = np.stack([zoom(image[d], (patch_size[0] / X, patch_size[1] / Y), order=3) if resize else image[d] for d in range(D)])
        pred = predict_slices(model, torch.from_numpy(sl.astype(np.float32)).to(dev), batch).cpu().numpy()
        if resize:
            pred = np.stack([zoom(pred[d], (X / patch_size[0], Y / patch_size[1]), order=0) for d in range(D)])
    else:
        vol = torch.from_numpy(np.ascontiguousarray(image, np.float32)).to(dev)
        sl = zoom_volume_to_network(vol, tuple(patch_size)) if resize else vol
        pred_d = predict_slices(model, sl, batch)
        pred = (zoom_labels(pred_d, (X, Y)) if resize else pred_d).cpu().numpy()
    if with_hd95:
        return [calculate_metric_percase(pred == k, label == k) for k in range(1, classes)]
    counts = np.zeros((classes, 3), dtype=np.float64)
    for k in range(classes):
        p, g = pred == k, label == k
        counts[k] = (np.logical_and(p, g).sum(), p.sum(), g.sum())
    return dice_from_counts(counts)


def inference(model, volumes, classes: int = 9, img_size: int = 224, batch: int = 16, log=None) -> Tuple[float, float]:
    """trainer.py:25-47: mean Dice and mean HD95 over `volumes` = iterable of (image [D,H,W] in [0,1], label [D,H,W], case name)."""
    total, n = 0.0, 0
    for i, (image, label, name) in enumerate(volumes):
        m = np.array(evaluate_volume(model, np.asarray(image), np.asarray(label), classes, (img_size, img_size), batch, with_hd95=True))
        total = total + m
        n += 1
        if log:
            log(' idx %d case %s mean_dice %f mean_hd95 %f' % (i, name, m.mean(axis=0)[0], m.mean(axis=0)[1]))
    if n == 0:
        raise ValueError("inference() needs at least one volume")
    total = total / n
    if log:
        for k in range(1, classes):
            log('Mean class %d mean_dice %f mean_hd95 %f' % (k, total[k - 1][0], total[k - 1][1]))
    performance, mean_hd95 = float(total.mean(axis=0)[0]), float(total.mean(axis=0)[1])
    if log:
        log('Testing performance in best val model: mean_dice : %f mean_hd95 : %f' % (performance, mean_hd95))
    return performance, mean_hd95
