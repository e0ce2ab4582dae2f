"""Input pipeline of the TransCeption path on MI355X: the device side of
`datasets/dataset_synapse.py::Synapse_dataset.__getitem__` (:101-128) and the trainer's transforms (`trainer.py:89-108`).

The reference prepares every training slice on host cores: `np.load` of `{image,label}` (:104-107), an imgaug `SomeOf((0,4))`
pipeline of ten augmenters (:84-95), `scipy.ndimage.zoom` order 3 / order 0 from 512x512 to the network size (:108-112), then
`ToTensor` + `Normalize([0.5],[0.5])` (trainer.py:89-93) -- about 40 ms per slice and core, i.e. ~25 slices/s/core against a GPU
that consumes ~850 slices/s.  Here the host only reads the npz files and draws the augmentation parameters; the raw slices go
to HBM as they are (fp32 image, uint8 label: 1.25 MB per slice) and the arithmetic runs there, four launches per batch
(`csrc/data.hip`): augment -> spline prefilter (columns, rows) -> resize + normalise.  The loader issues that work on its own
stream while the training step of the previous batch is running.

What is bit-for-bit the reference's arithmetic and what is not is spelled out in `oracle/data_oracle.py`: the resize/normalise
stage reproduces `scipy.ndimage.zoom` (including its zeroed last row/column at 512 -> 224) and is pinned by a fixture generated
from the reference's own code; the imgaug stage is a restatement (imgaug is not installable here), with the geometric
augmenters folded into ONE order-1 warp instead of one resampling per augmenter.
"""
from __future__ import annotations

import ctypes
import math
import os
import queue
import threading
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
from scipy import special

from ._lib import TC_AUG_BLUR, TC_AUG_FROM_RAW, TC_AUG_LINEAR, TC_AUG_PIECEWISE, TC_AUG_SKIP, TC_AUG_WARP, TcSliceAug, lib

NOISE_SCALE = 0.005 * 255          # AdditiveGaussianNoise(scale=0.005*255), dataset_synapse.py:87 -- applied to [0,1] floats as is


# ------------------------------------------------------------------------------------------------ affine maps (output -> source)
# A map is 6 numbers m: source_row = m[2] + m[0]*y + m[1]*x ; source_col = m[5] + m[3]*y + m[4]*x for output pixel (y, x).
def _to3(m) -> np.ndarray:
    return np.array([[m[0], m[1], m[2]], [m[3], m[4], m[5]], [0.0, 0.0, 1.0]], np.float64)


def _from3(a: np.ndarray) -> Tuple[float, ...]:
    return (float(a[0, 0]), float(a[0, 1]), float(a[0, 2]), float(a[1, 0]), float(a[1, 1]), float(a[1, 2]))


IDENTITY = (1.0, 0.0, 0.0, 0.0, 1.0, 0.0)


def compose(first, then) -> Tuple[float, ...]:
    """Map of the slice produced by applying augmentation `first` and then `then` (each given as its output->source map)."""
    return _from3(_to3(first) @ _to3(then))


def affine_flip(axis: int, h: int, w: int) -> Tuple[float, ...]:
    return (-1.0, 0.0, float(h - 1), 0.0, 1.0, 0.0) if axis == 0 else (1.0, 0.0, 0.0, 0.0, -1.0, float(w - 1))


def affine_rot90_flip(k: int, axis: int, n: int) -> Tuple[float, ...]:
    """np.flip(np.rot90(a, k), axis) of a square slice (random_rot_flip, dataset_synapse.py:39-46)."""
    rot = {0: IDENTITY, 1: (0.0, 1.0, 0.0, -1.0, 0.0, float(n - 1)), 2: (-1.0, 0.0, float(n - 1), 0.0, -1.0, float(n - 1)),
           3: (0.0, -1.0, float(n - 1), 1.0, 0.0, 0.0)}[k % 4]
    return compose(rot, affine_flip(axis, n, n))


def affine_rotate(angle_deg: float, h: int, w: int) -> Tuple[float, ...]:
    """scipy.ndimage.rotate(a, angle, reshape=False) (random_rotate, dataset_synapse.py:48-52): matrix [[c, s], [-s, c]] in
    (row, col), about the slice centre; cos/sin of degrees through scipy.special like scipy itself."""
    c, s = float(special.cosdg(angle_deg)), float(special.sindg(angle_deg))
    cy, cx = (h - 1) / 2.0, (w - 1) / 2.0
    return (c, s, cy - (c * cy + s * cx), -s, c, cx - (-s * cy + c * cx))


def _xy_forward_to_map(fwd_xy: np.ndarray, h: int, w: int) -> Tuple[float, ...]:
    """Forward transform in (x, y) pixel coordinates about the slice centre -> output->source map in (row, col)."""
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    t_in = np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1]], np.float64)
    t_out = np.array([[1, 0, cx], [0, 1, cy], [0, 0, 1]], np.float64)
    inv = np.linalg.inv(t_out @ fwd_xy @ t_in)                     # output (x, y) -> source (x, y)
    swap = np.array([[0, 1, 0], [1, 0, 0], [0, 0, 1]], np.float64)
    return _from3(swap @ inv @ swap)


def affine_scale(sx: float, sy: float, h: int, w: int):
    return _xy_forward_to_map(np.diag([sx, sy, 1.0]), h, w)


def affine_rotate_xy(deg: float, h: int, w: int):
    r = math.radians(deg)
    return _xy_forward_to_map(np.array([[math.cos(r), -math.sin(r), 0], [math.sin(r), math.cos(r), 0], [0, 0, 1]], np.float64), h, w)


def affine_shear(deg: float, h: int, w: int):
    r = math.radians(deg)
    return _xy_forward_to_map(np.array([[1, -math.sin(r), 0], [0, math.cos(r), 0], [0, 0, 1]], np.float64), h, w)


def affine_translate(px: float, py: float, h: int, w: int):
    return _xy_forward_to_map(np.array([[1, 0, px * w], [0, 1, py * h], [0, 0, 1]], np.float64), h, w)


MAX_ROUNDS = 4                             # SomeOf((0, 4), ...): at most four augmenters per slice (dataset_synapse.py:84)


_PW_DIAG: dict = {}


def piecewise_grid(h: int, w: int) -> np.ndarray:
    """imgaug 0.4.0 PiecewiseAffine control points: linspace(0, h, 4) x linspace(0, w, 4), (row, col), row-major."""
    xx, yy = np.meshgrid(np.linspace(0, w, 4), np.linspace(0, h, 4))
    return np.dstack([yy.flat, xx.flat])[0]


def piecewise_disp(jitter: np.ndarray, h: int, w: int) -> np.ndarray:
    """Control-point displacements [4,4,2] (dy, dx) of a drawn jitter: imgaug clips the moved points into the image plane."""
    src = piecewise_grid(h, w)
    dst = src + np.asarray(jitter, np.float64).reshape(-1, 2)
    dst[:, 0], dst[:, 1] = np.clip(dst[:, 0], 0, h - 1), np.clip(dst[:, 1], 0, w - 1)
    return (dst - src).astype(np.float32).reshape(4, 4, 2)


def piecewise_diagonals(h: int, w: int) -> int:
    """Nine bits, one per cell (gy * 3 + gx) of the 4x4 control grid: set when the Delaunay triangulation of the grid -- what skimage's
    PiecewiseAffineTransform builds from the source points (scipy.spatial.Delaunay) -- splits the cell along its top-right / bottom-left
    diagonal, clear for top-left / bottom-right.  The grid is regular, every cell's corners are co-circular, the choice is Qhull's."""
    key = (h, w)
    if key not in _PW_DIAG:
        from scipy import spatial
        tri = spatial.Delaunay(piecewise_grid(h, w)[:, ::-1])
        edges = {frozenset((int(a), int(b))) for t in tri.simplices for a, b in ((t[0], t[1]), (t[1], t[2]), (t[0], t[2]))}
        bits = 0
        for gy in range(3):
            for gx in range(3):
                tl = gy * 4 + gx
                a, b = frozenset((tl, tl + 5)) in edges, frozenset((tl + 1, tl + 4)) in edges
                assert a != b, "every cell of the control grid is split by exactly one diagonal"
                if b:
                    bits |= 1 << (gy * 3 + gx)
        _PW_DIAG[key] = bits
    return _PW_DIAG[key]


@dataclass
class SliceAugmentation:
    """Parameters of one slice's augmentation.  Either ONE stage (one TcSliceAug record: an output->source warp and / or pixel
    operations), or -- what AugmentSampler draws -- a chain of stages applied one after the other, each with its own resampling
    (`stages`: one single-stage SliceAugmentation per drawn augmenter, in the drawn order; imgaug resamples once per augmenter)."""
    m: Tuple[float, ...] = IDENTITY
    order: int = 1
    disp: Optional[np.ndarray] = None          # float32 [4,4,2] control-point displacement (dy, dx) in pixels (PiecewiseAffine: moved - source
    #                                            point of imgaug's 4x4 grid over [0, h] x [0, w], after its clip into the image: piecewise_disp)
    shape: Optional[Tuple[int, int]] = None    # (h, w) of the slices the record is for: required with `disp` (the grid's triangulation)
    blur: bool = False
    alpha: float = 1.0
    center: float = 0.0
    noise_sigma: float = 0.0
    noise_seed: int = 0
    names: List[str] = field(default_factory=list)
    pixel_order: Tuple[str, ...] = ()          # the pixel stages ("blur" / "contrast" / "noise") in the order they were drawn
    stages: Optional[List["SliceAugmentation"]] = None

    def rounds(self) -> List["SliceAugmentation"]:
        """The launches this slice needs: its chain, or itself as a single stage."""
        return list(self.stages) if self.stages is not None else [self]

    def order_code(self) -> int:
        """TcSliceAug.reserved: the drawn order of the pixel stages as 2-bit codes (0 = canonical blur -> contrast -> noise)."""
        code = 0
        for i, name in enumerate(self.pixel_order[:3]):
            code |= {"blur": 1, "contrast": 2, "noise": 3}[name] << (2 * i)
        return code

    def warps(self) -> bool:
        return self.disp is not None or tuple(self.m) != IDENTITY

    def record(self) -> TcSliceAug:
        r = TcSliceAug()
        for i, v in enumerate(self.m):
            r.m[i] = v
        flags = 0
        if self.warps():
            flags |= TC_AUG_WARP
        if self.order == 1:
            flags |= TC_AUG_LINEAR
        if self.disp is not None:
            assert self.shape is not None, "a piecewise warp needs the slice shape (the control grid's triangulation depends on it)"
            flags |= TC_AUG_PIECEWISE | (piecewise_diagonals(*self.shape) << 8)
            for i, v in enumerate(np.asarray(self.disp, np.float32).reshape(-1)):
                r.disp[i] = float(v)
        if self.blur:
            flags |= TC_AUG_BLUR
        r.alpha, r.center, r.noise_sigma, r.noise_seed, r.flags = self.alpha, self.center, self.noise_sigma, self.noise_seed, flags
        r.reserved = self.order_code()
        return r

    def as_dict(self) -> dict:
        """The same parameters in the form oracle/data_oracle.py::augment_slice takes (tests only)."""
        if self.stages is not None:
            return {"stages": [st.as_dict() for st in self.stages]}
        return {"m": tuple(self.m) if self.warps() else IDENTITY, "order": self.order,
                "disp": None if self.disp is None else np.asarray(self.disp, np.float32).reshape(-1),
                "blur": self.blur, "alpha": self.alpha, "center": self.center, "noise_sigma": self.noise_sigma,
                "noise_seed": self.noise_seed, "pixel_order": tuple(self.pixel_order)}


class AugmentSampler:
    """Draws what `iaa.SomeOf((0,4), [...ten augmenters...], random_order=True)` draws (dataset_synapse.py:84-95): between 0 and 4 of
    the ten augmenters, in random order, each with its own parameter ranges.  Every drawn augmenter is its own stage, applied in the
    drawn order with its own resampling (order 1 for the image, order 0 for the label, cval 0), as imgaug does: a later geometric
    augmenter sees what an earlier one moved out of frame as zeros, and every resampling softens the slice a little; the flips are
    exact copies.  Coordinates follow imgaug 0.4's pixel-centre convention (transforms about ((w - 1) / 2, (h - 1) / 2), its
    `_AffineMatrixGenerator` shift of size / 2 - 0.5).  Not restated: cv2.warpAffine's fixed-point coordinates (1/32-pixel
    interpolation tables).  PiecewiseAffine follows skimage's per-triangle PiecewiseAffineTransform over the Delaunay triangulation of
    imgaug's 4x4 control grid (piecewise_grid / piecewise_disp / piecewise_diagonals).  imgaug itself is not importable here, so this
    stage stays parity-unpinned (its arithmetic is defined by oracle/data_oracle.py)."""
    NAMES = ("Flipud", "Fliplr", "AdditiveGaussianNoise", "GaussianBlur", "LinearContrast", "Affine.scale", "Affine.rotate",
             "Affine.shear", "PiecewiseAffine", "Affine.translate")

    def __init__(self, seed: int):
        self.rng = np.random.default_rng(seed)

    def sample(self, h: int, w: int) -> SliceAugmentation:
        g = self.rng
        a = SliceAugmentation(stages=[])
        k = int(g.integers(0, 5))
        for idx in g.permutation(10)[:k]:
            name = self.NAMES[int(idx)]
            a.names.append(name)
            st = SliceAugmentation()
            if name == "Flipud":
                if g.random() < 0.5:
                    st.m = affine_flip(0, h, w)                  # (integer source coordinates: an exact copy, as imgaug's array flip)
            elif name == "Fliplr":
                if g.random() < 0.5:
                    st.m = affine_flip(1, h, w)
            elif name == "AdditiveGaussianNoise":
                st.noise_sigma, st.noise_seed = NOISE_SCALE, int(g.integers(0, 2 ** 31 - 1))
            elif name == "GaussianBlur":
                st.blur = True
            elif name == "LinearContrast":
                st.alpha = float(g.uniform(0.5, 1.5))
            elif name == "Affine.scale":
                st.m = affine_scale(float(g.uniform(0.5, 2.0)), float(g.uniform(0.5, 2.0)), h, w)
            elif name == "Affine.rotate":
                st.m = affine_rotate_xy(float(g.uniform(-40, 40)), h, w)
            elif name == "Affine.shear":
                st.m = affine_shear(float(g.uniform(-16, 16)), h, w)
            elif name == "PiecewiseAffine":
                sc = float(g.uniform(0.008, 0.03))
                st.disp, st.shape = piecewise_disp(g.normal(0.0, 1.0, (4, 4, 2)) * np.array([sc * h, sc * w]), h, w), (h, w)
            else:
                st.m = affine_translate(float(g.uniform(-0.2, 0.2)), float(g.uniform(-0.2, 0.2)), h, w)
            if st.warps() or st.blur or st.alpha != 1.0 or st.noise_sigma > 0.0:
                a.stages.append(st)
        return a


# ------------------------------------------------------------------------------------------------ files
def write_synthetic_synapse(base_dir: str, list_dir: str, n_cases: int = 2, slices_per_case: int = 8, size: int = 512,
                            seed: int = 1234) -> List[str]:
    """Writes a synthetic Synapse training set in the reference's on-disk format: `<base_dir>/caseNNNN_sliceMMM.npz` holding
    `image` float32 [size,size] in [0,1] and `label` float32 [size,size] in {0..8} (dataset_synapse.py:103-107), names listed in
    `<list_dir>/train.txt` (:80).  Smooth organ-like blobs, seeded."""
    os.makedirs(base_dir, exist_ok=True)
    os.makedirs(list_dir, exist_ok=True)
    g = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.linspace(-1, 1, size, dtype=np.float32), np.linspace(-1, 1, size, dtype=np.float32), indexing="ij")
    names = []
    for c in range(n_cases):
        centres = g.uniform(-0.6, 0.6, (8, 2)).astype(np.float32)
        radii = g.uniform(0.08, 0.3, (8, 2)).astype(np.float32)
        for s in range(slices_per_case):
            grow = np.float32(0.6 + 0.4 * math.sin(math.pi * (s + 0.5) / slices_per_case))
            image = 0.25 + 0.1 * np.sin(3 * xx + c) * np.cos(2 * yy - s * 0.1)
            label = np.zeros((size, size), np.float32)
            for k in range(8):
                d = ((yy - centres[k, 0]) / (radii[k, 0] * grow)) ** 2 + ((xx - centres[k, 1]) / (radii[k, 1] * grow)) ** 2
                inside = d < 1.0
                label[inside] = k + 1
                image = np.where(inside, 0.35 + 0.07 * k + 0.05 * (1 - d), image)
            image = np.clip(image + g.normal(0, 0.02, image.shape), 0, 1).astype(np.float32)
            name = f"case{c:04d}_slice{s:03d}"
            np.savez(os.path.join(base_dir, name + ".npz"), image=image, label=label)
            names.append(name)
    with open(os.path.join(list_dir, "train.txt"), "w") as f:
        f.write("\n".join(names) + "\n")
    return names


def _npz_member(f, zf, name: str):
    """(dtype, shape, data offset) of an uncompressed C-order .npy member of an open npz, or None (compressed / Fortran / pickled):
    np.savez stores members uncompressed, so a slice can be read from the file straight into its destination buffer."""
    import struct
    import zipfile
    try:
        info = zf.getinfo(name + ".npy")
    except KeyError:
        return None
    if info.compress_type != zipfile.ZIP_STORED:
        return None
    f.seek(info.header_offset)
    hdr = f.read(30)
    if len(hdr) < 30 or hdr[:4] != b"PK\x03\x04":
        return None
    nlen, elen = struct.unpack("<HH", hdr[26:30])
    start = info.header_offset + 30 + nlen + elen
    f.seek(start)
    magic = f.read(8)
    if magic[:6] != b"\x93NUMPY":
        return None
    major = magic[6]
    hlen = struct.unpack("<H", f.read(2))[0] if major == 1 else struct.unpack("<I", f.read(4))[0]
    import ast
    meta = ast.literal_eval(f.read(hlen).decode("latin1"))
    if meta.get("fortran_order") or not isinstance(meta.get("descr"), str):
        return None
    return np.dtype(meta["descr"]), tuple(meta["shape"]), start + 8 + (2 if major == 1 else 4) + hlen


def read_slice_into(path: str, image_out: np.ndarray, label_out: np.ndarray) -> None:
    """`np.load(path)['image' | 'label']` (dataset_synapse.py:104-107) written into caller buffers (float32 [H,W] / uint8 [H,W], e.g.
    views of pinned staging memory): uncompressed members are read from the file straight into place (no intermediate arrays, the
    GIL is released for the read), anything else goes through np.load."""
    import zipfile
    try:
        with open(path, "rb") as f, zipfile.ZipFile(f) as zf:
            mi, ml = _npz_member(f, zf, "image"), _npz_member(f, zf, "label")
            if mi is not None and ml is not None and mi[1] == image_out.shape and ml[1] == label_out.shape:
                if mi[0] == np.float32 and image_out.flags.c_contiguous:
                    f.seek(mi[2])
                    if f.readinto(memoryview(image_out).cast("B")) != image_out.nbytes:
                        raise IOError("short read")
                else:
                    f.seek(mi[2])
                    image_out[...] = np.fromfile(f, mi[0], image_out.size).reshape(mi[1])
                f.seek(ml[2])
                label_out[...] = np.fromfile(f, ml[0], label_out.size).reshape(ml[1])      # float32 0..8 -> uint8
                return
    except (zipfile.BadZipFile, ValueError, SyntaxError):
        pass
    data = np.load(path)
    image_out[...] = data["image"]
    label_out[...] = data["label"]


class SynapseSlices:
    """`Synapse_dataset(split="train")` without the transforms (dataset_synapse.py:75-82,101-107): slice i -> raw `image`
    (float32 [H,W] in [0,1]) and `label` (uint8 [H,W]; the file stores float32 0..8).  The test split (`.npy.h5` volumes, :114-118)
    needs h5py, which this image lacks -- volumes are handed to transception_amd.evaluate as arrays instead."""

    def __init__(self, base_dir: str, list_dir: str, split: str = "train"):
        if split != "train":
            raise NotImplementedError("only the npz training split is read here; evaluate_volume() takes volumes as arrays")
        path = os.path.join(list_dir, split + ".txt")
        with open(path) as f:
            self.sample_list = [ln.strip("\n") for ln in f.readlines() if ln.strip()]
        self.data_dir = base_dir

    def __len__(self) -> int:
        return len(self.sample_list)

    def __getitem__(self, idx: int):
        name = self.sample_list[idx]
        data = np.load(os.path.join(self.data_dir, name + ".npz"))
        image = np.ascontiguousarray(data["image"], np.float32)
        label = np.ascontiguousarray(data["label"]).astype(np.uint8)
        return image, label, name

    def shape(self, idx: int = 0):
        return self[idx][0].shape

    def read_into(self, idx: int, image_out: np.ndarray, label_out: np.ndarray) -> str:
        """Slice idx written into caller buffers (see read_slice_into); returns its name."""
        name = self.sample_list[idx]
        read_slice_into(os.path.join(self.data_dir, name + ".npz"), image_out, label_out)
        return name


# ------------------------------------------------------------------------------------------------ device side
def preprocess_batch(images: torch.Tensor, labels: torch.Tensor, augs: Optional[Sequence[Optional[SliceAugmentation]]], size: int,
                     mean: float = 0.5, std: float = 0.5, records: Optional[torch.Tensor] = None, scratch: Optional[dict] = None,
                     out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, rounds: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Raw slices in HBM -> network input, on the current stream.  images float32 [B,H,W], labels uint8 [B,H,W];
    `augs` one SliceAugmentation (or None) per slice, or None for no augmentation at all; alternatively `records` = the
    TcSliceAug array already on the device (uint8 [B, sizeof], or pack_rounds' [MAX_ROUNDS, B, sizeof] of which the LAST `rounds` are
    launched: one launch per round, each slice's stage resampling its previous stage's result -- tc_slice_augment_chain).  Returns x float32 [B,1,size,size], y int64
    [B,size,size]."""
    if not images.is_cuda:
        raise RuntimeError("transception_amd.data preprocesses on MI355X only (no CPU fallback)")
    B, H, W = images.shape
    assert images.dtype == torch.float32 and labels.dtype == torch.uint8 and labels.shape == images.shape
    images, labels = images.contiguous(), labels.contiguous()
    dev = images.device
    L = lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    scratch = scratch if scratch is not None else {}

    def buf(name, shape, dtype):
        t = scratch.get(name)
        if t is None or t.shape != torch.Size(shape) or t.dtype != dtype:
            t = scratch[name] = torch.empty(shape, dtype=dtype, device=dev)
        return t

    if records is None and augs is not None and any(a is not None for a in augs):
        rec_np, rounds = pack_rounds(augs)
        records = torch.from_numpy(rec_np).to(dev)
    if records is not None and records.dim() == 2:                # one single-stage record per slice
        img2, lab2 = buf("aug_img1", (B, H, W), torch.float32), buf("aug_lab1", (B, H, W), torch.uint8)
        L.tc_slice_augment(images.data_ptr(), labels.data_ptr(), records.data_ptr(), img2.data_ptr(), lab2.data_ptr(), B, H, W, stream)
        images, labels = img2, lab2
    elif records is not None:                                     # chains (pack_rounds): the last `rounds` of the MAX_ROUNDS rounds
        R = int(records.shape[0])
        n = R if rounds is None else int(rounds)
        if n > 0:
            bufs = [buf(f"aug_img{i}", (B, H, W), torch.float32) for i in (0, 1)] + [buf(f"aug_lab{i}", (B, H, W), torch.uint8) for i in (0, 1)]
            L.tc_slice_augment_chain(images.data_ptr(), labels.data_ptr(), records.data_ptr(), R - n, R, bufs[0].data_ptr(), bufs[2].data_ptr(),
                                     bufs[1].data_ptr(), bufs[3].data_ptr(), B, H, W, stream)
            images, labels = bufs[(R - 1) & 1], bufs[2 + ((R - 1) & 1)]
    x, y = out if out is not None else (torch.empty((B, 1, size, size), dtype=torch.float32, device=dev),
                                        torch.empty((B, size, size), dtype=torch.int64, device=dev))
    if H != size or W != size:
        coef = buf("coef", (B, H, W), torch.float64)
        L.tc_spline_prefilter(images.data_ptr(), coef.data_ptr(), B, H, W, stream)
        L.tc_zoom_normalize(coef.data_ptr(), images.data_ptr(), labels.data_ptr(), x.data_ptr(), y.data_ptr(), B, H, W, size, size,
                            float(mean), float(std), stream)
    else:
        L.tc_zoom_normalize(None, images.data_ptr(), labels.data_ptr(), x.data_ptr(), y.data_ptr(), B, H, W, size, size,
                            float(mean), float(std), stream)
    return x, y


def pack_records(augs: Sequence[Optional[SliceAugmentation]]) -> np.ndarray:
    """TcSliceAug array as bytes, uint8 [B, sizeof(TcSliceAug)] (single-stage augmentations)."""
    assert all(a is None or a.stages is None for a in augs), "chains of stages go through pack_rounds"
    arr = (TcSliceAug * len(augs))(*[(a if a is not None else SliceAugmentation()).record() for a in augs])
    return np.frombuffer(bytes(arr), np.uint8).reshape(len(augs), ctypes.sizeof(TcSliceAug)).copy()


def pack_rounds(augs: Sequence[Optional[SliceAugmentation]]) -> Tuple[np.ndarray, int]:
    """(uint8 [MAX_ROUNDS, B, sizeof(TcSliceAug)], n = the longest chain) for tc_slice_augment_chain.  Chains END in the last round: stage
    j of a chain of k stages sits in round MAX_ROUNDS - k + j, its first stage marked TC_AUG_FROM_RAW (it reads the raw slice); the rounds
    before a chain starts hold TC_AUG_SKIP records (nothing is read or written: no identity copies between the ping-pong buffers), and a
    slice without stages is copied once, in the last round (identity record | TC_AUG_FROM_RAW).  Every slice's result is then in the
    buffer the last round writes, whichever of rounds [MAX_ROUNDS - n, MAX_ROUNDS) -- or all of them, in a captured step -- are launched."""
    chains = [(a.rounds() if a is not None else []) for a in augs]
    n = max([len(c) for c in chains] + [0])
    assert n <= MAX_ROUNDS
    sz = ctypes.sizeof(TcSliceAug)
    arr = ((TcSliceAug * len(augs)) * MAX_ROUNDS)()
    for b, c in enumerate(chains):
        k = len(c)
        for r in range(MAX_ROUNDS):
            j = r - (MAX_ROUNDS - k)
            if j >= 0:
                rec = c[j].record()
                if j == 0:
                    rec.flags |= TC_AUG_FROM_RAW
            else:
                rec = SliceAugmentation().record()
                rec.flags |= (TC_AUG_FROM_RAW if (k == 0 and r == MAX_ROUNDS - 1) else TC_AUG_SKIP)
            arr[r][b] = rec
    return np.frombuffer(bytes(arr), np.uint8).reshape(MAX_ROUNDS, len(augs), sz).copy(), n


def epoch_order(n: int, epoch: int, seed: int, shuffle: bool = True) -> np.ndarray:
    """Slice order of one epoch -- the same on every rank (the trainer's DataLoader(shuffle=True), trainer.py:104)."""
    return np.random.default_rng([seed, epoch]).permutation(n) if shuffle else np.arange(n)


def rank_batches(order: np.ndarray, batch_size: int, rank: int, world: int) -> List[np.ndarray]:
    """Global batches of batch_size*world slices (trainer.py:86), rank r taking slices [r*B, (r+1)*B) of each.  ceil(N / global
    batch) batches per epoch, like the reference's DataLoader (drop_last=False, trainer.py:104: 93 iterations for Synapse's 2211
    slices at B=24, and a cosine T_max of max_epochs * 93); the captured step needs full batches, so the last one is completed
    from the head of the same permutation instead of being short.  Stated deviation: the reference's last batch of an epoch holds
    N mod global-batch slices (3 of 24 for Synapse) whose loss is a mean over 3; here those slices share a full batch with
    global-batch - 3 slices that the epoch has already seen (each of them weighs 1/24, and 21 slices are seen twice per epoch)."""
    gb = batch_size * world
    n = len(order)
    if n == 0:
        return []
    nb = -(-n // gb)
    padded = np.concatenate([order, order[:nb * gb - n]]) if nb * gb > n else order
    while len(padded) < nb * gb:                               # data sets smaller than one global batch: keep wrapping
        padded = np.concatenate([padded, order[:nb * gb - len(padded)]])
    return [padded[i * gb + rank * batch_size: i * gb + (rank + 1) * batch_size] for i in range(nb)]


_DEBUG_SKIP_H2D = bool(os.environ.get("TC_LOADER_SKIP_H2D"))
_DEBUG_NO_EVENTS = bool(os.environ.get("TC_LOADER_NO_EVENTS"))      # timing what-if only (racy)
_HOST_WAITS_FOR_COPY = os.environ.get("TC_LOADER_HOST_WAIT", "1") != "0"


class DeviceLoader:
    """Iterates (x, y) batches resident in HBM.  A host thread (with a pool of readers) fills pinned staging buffers from the npz
    files and draws the augmentation parameters; host-to-device copies run two batches ahead on the loader's stream; the four
    preprocessing launches of a batch are issued on the consumer's stream just before the batch is handed out."""

    def __init__(self, dataset: SynapseSlices, batch_size: int, img_size: int = 224, device="cuda", seed: int = 1234, rank: int = 0,
                 world: int = 1, augment: bool = True, shuffle: bool = True, epochs: int = 1, prefetch: int = 3, readers: int = 8,
                 out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        self.ds, self.B, self.size, self.device = dataset, batch_size, img_size, torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceLoader feeds an MI355X; there is no CPU path")
        self.seed, self.rank, self.world, self.augment, self.shuffle, self.epochs = seed, rank, world, augment, shuffle, epochs
        self.stream = torch.cuda.Stream(self.device)
        self.prefetch, self.readers = prefetch, max(1, readers)
        # out = (x float32 [B,1,S,S], y int64 [B,S,S]): every batch is written there (e.g. the static inputs of a captured step: no
        # device-to-device copy at the step boundary, where it would queue behind the loader's host-to-device transfer)
        self.out = out
        self.q: "queue.Queue" = queue.Queue(maxsize=prefetch)
        self.free: "queue.Queue" = queue.Queue()                     # pinned staging sets handed back once their copies ran
        self._staged = 0
        self._hw = None                                              # slice size (one per dataset, as in Synapse)
        self.slots = [dict(scratch={}) for _ in range(3)]
        self._thread: Optional[threading.Thread] = None
        self._stop = threading.Event()

    def __len__(self) -> int:
        return -(-len(self.ds) // (self.B * self.world)) * self.epochs        # ceil, like DataLoader(drop_last=False): see rank_batches

    # host side -----------------------------------------------------------------------------------
    def _staging(self, h: int, w: int):
        """A pinned (image, label, records) set: at most prefetch + 4 exist (queue + the three device slots + one being filled);
        page-locking memory per batch would cost more than reading the slices."""
        while not self._stop.is_set():
            try:
                st = self.free.get_nowait() if self._staged >= self.prefetch + 4 else None
            except queue.Empty:
                try:
                    st = self.free.get(timeout=0.1)
                except queue.Empty:
                    continue
            if st is None:
                self._staged += 1
                return (torch.empty((self.B, h, w), dtype=torch.float32).pin_memory(), torch.empty((self.B, h, w), dtype=torch.uint8).pin_memory(),
                        torch.empty((MAX_ROUNDS, self.B, ctypes.sizeof(TcSliceAug)), dtype=torch.uint8).pin_memory())
            if st[0].shape[1:] == (h, w):
                return st
            self._staged -= 1                                        # slice size changed: drop the set, make a new one
        return None

    def _produce(self):
        from concurrent.futures import ThreadPoolExecutor
        # CPython hands the GIL over every 5 ms by default: while this thread packs augmentation records, the consumer's next graph
        # launch would wait up to that long (measured: 0.3-1.7 ms idle gaps at step boundaries of a 12.6 ms step).  A short switch
        # interval lets the launching thread in at once; restored when the producer ends.
        import sys as _sys
        old_iv = _sys.getswitchinterval()
        _sys.setswitchinterval(min(old_iv, float(os.environ.get("TC_LOADER_SWITCH_INTERVAL", "2e-5"))))
        try:
            self._produce_body(ThreadPoolExecutor)
        finally:
            _sys.setswitchinterval(old_iv)

    def _produce_body(self, ThreadPoolExecutor):
        try:
            with ThreadPoolExecutor(max_workers=self.readers) as pool:
                for epoch in range(self.epochs):
                    sampler = AugmentSampler(int(np.random.SeedSequence([self.seed, epoch, self.rank]).generate_state(1)[0]))
                    for idxs in rank_batches(epoch_order(len(self.ds), epoch, self.seed, self.shuffle), self.B, self.rank, self.world):
                        if self._hw is None:
                            self._hw = tuple(self.ds.shape(int(idxs[0])))
                        h, w = self._hw
                        st = self._staging(h, w)
                        if st is None:
                            return
                        img, lab, rec = st
                        img_np, lab_np = img.numpy(), lab.numpy()           # views of the pinned staging memory

                        def fill(j):
                            return self.ds.read_into(int(idxs[j]), img_np[j], lab_np[j])   # raises on a slice of another size
                        names = list(pool.map(fill, range(len(idxs))))
                        augs = [sampler.sample(h, w) for _ in names] if self.augment else None
                        nr = 0
                        if augs is not None:
                            rec_np, nr = pack_rounds(augs)
                            rec.copy_(torch.from_numpy(rec_np))
                        while not self._stop.is_set():
                            try:
                                self.q.put((st, names, augs, nr), timeout=0.1)
                                break
                            except queue.Full:
                                continue
                        if self._stop.is_set():
                            return
            self.q.put(None)
        except BaseException as e:                                   # surfaces in the consumer
            self.q.put(e)

    # device side ---------------------------------------------------------------------------------
    def _copy(self, slot: dict) -> bool:
        """Next staged batch -> the slot's raw device buffers, on the loader's stream (copy engine, under the running step)."""
        item = self.q.get()
        if item is None:
            return False
        if isinstance(item, BaseException):
            raise item
        st, names, augs, nr = item
        img, lab, rec = st
        if "host" in slot:
            slot["copied"].synchronize()                             # that copy ran long ago: hand its staging set back
            self.free.put(slot.pop("host"))
        with torch.cuda.stream(self.stream):
            if "raw" not in slot or slot["raw"][0].shape != img.shape or (self.out is not None and slot["out"] is not self.out):
                slot["raw"] = (torch.empty(img.shape, dtype=torch.float32, device=self.device),
                               torch.empty(lab.shape, dtype=torch.uint8, device=self.device),
                               torch.empty(rec.shape, dtype=torch.uint8, device=self.device))
                slot["out"] = self.out if self.out is not None else (
                    torch.empty((self.B, 1, self.size, self.size), dtype=torch.float32, device=self.device),
                    torch.empty((self.B, self.size, self.size), dtype=torch.int64, device=self.device))
            if "prepped" in slot and not os.environ.get("TC_LOADER_NO_STREAM_WAIT"):
                # The kernels that read the slot's previous contents are done.  The HOST waits (the callers arrange that this event
                # is at least one step old, so the wait ends while the GPU still has a queued step to run); letting the loader's
                # stream wait for it on the device costs the CONSUMER's stream ~0.55 ms of idle time per step on this runtime when the
                # event sits behind a graph launch (13.34 vs 12.79 ms per loader-fed step, copies disabled in both).
                if _HOST_WAITS_FOR_COPY:
                    slot["prepped"].synchronize()
                else:
                    self.stream.wait_event(slot["prepped"])
            if not _DEBUG_SKIP_H2D or "h2d_once" not in slot:        # (timing what-if: TC_LOADER_SKIP_H2D=1 reuses the slot's first batch)
                slot["raw"][0].copy_(img, non_blocking=True)
                slot["raw"][1].copy_(lab, non_blocking=True)
                if augs is not None:
                    slot["raw"][2].copy_(rec, non_blocking=True)
                slot["h2d_once"] = True
            slot["copied"] = torch.cuda.Event()
            slot["copied"].record(self.stream)
        slot["host"], slot["names"], slot["augs"], slot["rounds"] = st, names, augs, nr
        return True

    def _wait_copied(self, slot: dict, main):
        """The batch in `slot` is on the device before anything the consumer launches reads it.  The HOST waits for the copy's event
        (issued one or two steps ago on the loader's stream: it has long completed) instead of making the consumer's stream wait for
        it: a cross-stream wait in front of a graph launch costs ~0.55 ms of idle GPU per step on this runtime (measured: 13.36 vs
        12.80 ms per loader-fed step), an event record behind it does not."""
        if _HOST_WAITS_FOR_COPY:
            slot["copied"].synchronize()
        else:
            main.wait_event(slot["copied"])

    def _prep(self, slot: dict):
        """The four preprocessing launches on the CONSUMER's stream, right in front of the step that uses the batch (kernels of
        different streams do not overlap on this GPU anyway -- see DESIGN.md -- and a side stream's launches slowed the step's)."""
        main = torch.cuda.current_stream(self.device)
        self._wait_copied(slot, main)
        d_img, d_lab, d_rec = slot["raw"]
        slot["x"], slot["y"] = preprocess_batch(d_img, d_lab, None, self.size, records=d_rec if (slot["augs"] is not None and slot["rounds"]) else None,
                                                scratch=slot["scratch"], out=slot["out"], rounds=slot["rounds"])
        slot["prepped"] = torch.cuda.Event()
        slot["prepped"].record(main)

    def iter_raw(self, nslots: int = 3):
        """Yields device slots holding one RAW batch each -- {"index", "raw": (float32 [B,H,W], uint8 [B,H,W], TcSliceAug records uint8
        [MAX_ROUNDS,B,sizeof] or None)} -- for a consumer that runs the preprocessing itself, e.g. captured at the head of its step
        graph (`slot_preprocess(slot)` below gives the callable; train.GraphedStep(pre=...)): one captured step per slot, `nslots`
        of them, the host-to-device copy of the batch after next running under the current step.  The slot's buffers are static (same
        addresses for the whole iteration).  The consumer must have issued everything that reads the slot on the current stream
        before it asks for the next one.  A slot is refilled one step AFTER the step that read it was issued (the host then waits
        for that step's event while the next step is already queued), hence three slots for an uninterrupted GPU."""
        from collections import deque
        assert 1 <= nslots <= len(self.slots)
        self._stop.clear()
        self._thread = threading.Thread(target=self._produce, daemon=True)
        self._thread.start()
        slots = self.slots[:nslots]
        for i, sl in enumerate(slots):
            sl["index"] = i
        free, copied, busy = deque(slots), deque(), deque()
        eof = False

        def try_copy():
            nonlocal eof
            if eof or not free:
                return
            slot = free.popleft()
            if self._copy(slot):
                copied.append(slot)
            else:
                eof = True
                free.appendleft(slot)
        try:
            for _ in range(max(1, nslots - 1)):
                try_copy()
            while copied:
                slot = copied.popleft()
                main = torch.cuda.current_stream(self.device)
                if not _DEBUG_NO_EVENTS:
                    self._wait_copied(slot, main)
                self.last_names, self.last_augs = slot["names"], slot["augs"]
                yield slot
                if not _DEBUG_NO_EVENTS:
                    slot["prepped"] = torch.cuda.Event()             # everything the consumer issued that reads the slot is in front of this
                    slot["prepped"].record(torch.cuda.current_stream(self.device))
                busy.append(slot)
                while len(busy) > (1 if nslots > 2 else 0):          # the slot of the step BEFORE the one just issued is refilled now
                    free.append(busy.popleft())
                try_copy()
        finally:
            self._stop.set()
            while self._thread.is_alive():
                try:
                    self.q.get_nowait()
                except queue.Empty:
                    pass
                self._thread.join(timeout=0.05)
            while not self.q.empty():
                self.q.get_nowait()

    def slot_preprocess(self, slot: dict):
        """pre(x, y) for train.GraphedStep: the preprocessing launches of the batch in `slot` (always MAX_ROUNDS augmentation rounds --
        the count must not depend on the batch inside a captured graph; a slice's chain ends in the last round and the rounds before it
        starts hold skip records: pack_rounds), writing the network input into x / y."""
        d_img, d_lab, d_rec = slot["raw"]
        scratch, size, aug = slot["scratch"], self.size, self.augment

        def pre(x, y):
            preprocess_batch(d_img, d_lab, None, size, records=d_rec if aug else None, scratch=scratch, out=(x, y), rounds=MAX_ROUNDS if aug else None)
        return pre

    def __iter__(self):
        from collections import deque
        self._stop.clear()
        self._thread = threading.Thread(target=self._produce, daemon=True)
        self._thread.start()
        free, copied = deque(self.slots), deque()
        eof = False

        def try_copy():
            nonlocal eof
            if eof or not free:
                return
            slot = free.popleft()
            if self._copy(slot):
                copied.append(slot)
            else:
                eof = True
                free.appendleft(slot)
        try:
            try_copy()
            try_copy()                                               # two host-to-device copies ahead of the consumer
            while copied:
                slot = copied.popleft()
                self._prep(slot)
                try_copy()                                           # goes out now, runs under the step that consumes `slot`
                self.last_names, self.last_augs = slot["names"], slot["augs"]      # what the batch being handed out was made from
                yield slot["x"], slot["y"]
                free.append(slot)                                    # its outputs were consumed in stream order
        finally:
            self._stop.set()
            while self._thread.is_alive():                           # unblock a producer waiting on a full queue
                try:
                    self.q.get_nowait()
                except queue.Empty:
                    pass
                self._thread.join(timeout=0.05)
            while not self.q.empty():
                self.q.get_nowait()
