"""CPU restatement of the reference's per-slice input arithmetic.  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and nothing in the product (transception_amd/ never imports oracle/).

What the reference does to one training slice (datasets/dataset_synapse.py:101-128, trainer.py:89-96):

    npz{image f32 [512,512] in [0,1], label f32 [512,512] in 0..8}  (dataset_synapse.py:104-107)
      -> imgaug SomeOf((0,4), 10 augmenters, random_order)           (:84-95, augment_seg :27-37)
      -> scipy.ndimage.zoom(image, 224/512, order=3), zoom(label, order=0)   (:108-112)
      -> ToTensor + Normalize([0.5],[0.5]) on the image, ToTensor on the label (trainer.py:89-93)

and, in the older generator kept beside it (RandomGenerator :56-73): rot90+flip (:39-46) or ndimage.rotate(order=0,
reshape=False) (:48-52) before the same zoom.

Pinning:
  * resize + normalise: the arithmetic lives in scipy.ndimage (third party; this image has scipy 1.15.3, the reference pins
    scipy through its requirements).  `resize_normalize` calls the very function the reference calls, and
    `zoom_cubic_restated` / `zoom_nearest_restated` restate its published algorithm (cubic B-spline prefilter with mirror
    boundaries, pole sqrt(3)-2, ni_splines.c; 4x4-tap evaluation with mirrored tap indices; coordinate o*(in-1)/(out-1) in
    double, and -- a quirk that matters -- an output whose coordinate rounds ABOVE in-1 gets cval=0: at 512->224 the whole last
    row and column of image and label are zero) -- tests check the restatement against scipy and against
    tests/golden/data_pipeline.npz, which was produced by running the reference's own RandomGenerator.
  * rot90/flip and order-0 rotate: pinned by the same fixture.
  * the imgaug family (Flipud/Fliplr/AdditiveGaussianNoise/GaussianBlur/LinearContrast/Affine x4/PiecewiseAffine): imgaug
    (requirements: imgaug 0.4.0) is NOT in this image and cannot be imported, so that stage is "parity unpinned": `augment_slice`
    below defines the arithmetic the HIP kernel has to match, it is not checked against imgaug.  What follows imgaug 0.4's
    published algorithm: the drawn augmenters are applied ONE AFTER THE OTHER in the drawn order (`SomeOf(random_order=True)`),
    every geometric one with its own order-1 resampling of the image (order 0 for the segmentation map) and cval 0
    (`Affine(order=1, cval=0, mode="constant")` defaults -> cv2.warpAffine INTER_LINEAR / BORDER_CONSTANT); transforms are taken
    about the pixel-centre midpoint ((w - 1) / 2, (h - 1) / 2) (imgaug's `_AffineMatrixGenerator`: shift = size / 2 - 0.5); the
    flips are array flips (exact); PiecewiseAffine is skimage's per-triangle `PiecewiseAffineTransform` over the Delaunay triangulation
    of imgaug's control grid (`piecewise_source`).  What is NOT restated: cv2.warpAffine's fixed-point source coordinates
    (interpolation tables of 1/32 pixel); and which diagonal Qhull picks in the (co-circular) cells of the regular grid is whatever
    THIS image's scipy picks -- the pinned scipy 1.5.4 may pick differently.
"""
import math

import numpy as np
from scipy import ndimage

SPLINE_POLE = math.sqrt(3.0) - 2.0
BLUR_RADIUS = 4                       # GaussianBlur(sigma=1.0) truncated at 4 sigma


# ------------------------------------------------------------------------------------------------ resize + normalise
def resize_normalize(image: np.ndarray, label: np.ndarray, size: int, mean: float = 0.5, std: float = 0.5):
    """dataset_synapse.py:108-112 + trainer.py:89-93.  Returns (x float32 [1,size,size], y int64 [size,size])."""
    h, w = image.shape
    if h != size or w != size:
        image = ndimage.zoom(image, (size / h, size / w), order=3)
        label = ndimage.zoom(label, (size / h, size / w), order=0)
    x = (image.astype(np.float32)[None] - np.float32(mean)) / np.float32(std)
    return x, label.astype(np.int64)


def spline_prefilter_line(c: np.ndarray) -> np.ndarray:
    """Cubic B-spline coefficients of one line, mirror boundary (scipy ni_splines.c: gain 6, exact causal initialisation,
    causal pass c[i] += z c[i-1], anti-causal initialisation, anti-causal pass c[i] = z (c[i+1] - c[i]))."""
    z = SPLINE_POLE
    n = c.shape[0]
    c = c.astype(np.float64) * 6.0
    if n == 1:
        return c / 6.0
    zn1 = z ** (n - 1)
    c0, zi = c[0] + zn1 * c[n - 1], z
    for i in range(1, n - 1):
        c0 += zi * (c[i] + zn1 * c[n - 1 - i])
        zi *= z
    c[0] = c0 / (1.0 - zn1 * zn1)
    for i in range(1, n):
        c[i] += z * c[i - 1]
    c[n - 1] = (z * c[n - 2] + c[n - 1]) * z / (z * z - 1.0)
    for i in range(n - 2, -1, -1):
        c[i] = z * (c[i + 1] - c[i])
    return c


def spline_prefilter(a: np.ndarray) -> np.ndarray:
    c = a.astype(np.float64)
    c = np.stack([spline_prefilter_line(c[:, j]) for j in range(c.shape[1])], 1)      # axis 0 first, as scipy does
    return np.stack([spline_prefilter_line(c[i, :]) for i in range(c.shape[0])], 0)


def _mirror(i: int, n: int) -> int:
    if n == 1:
        return 0
    p = 2 * (n - 1)
    i = abs(i) % p
    return p - i if i >= n else i


def _cubic_taps(coord: float):
    f = math.floor(coord)
    y = coord - f
    z = 1.0 - y
    w1 = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0
    w2 = (z * z * (z - 2.0) * 3.0 + 4.0) / 6.0
    w0 = z * z * z / 6.0
    return int(f) - 1, (w0, w1, w2, 1.0 - w0 - w1 - w2)


def zoom_coords(n_in: int, n_out: int):
    """Source coordinate of every output index and whether scipy treats it as outside (-> cval 0)."""
    step = (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
    cc = np.arange(n_out, dtype=np.float64) * step
    return cc, cc > (n_in - 1)


def zoom_cubic_restated(a: np.ndarray, oh: int, ow: int) -> np.ndarray:
    """scipy.ndimage.zoom(a, order=3, mode='constant', prefilter=True, grid_mode=False), small inputs only (Python loops)."""
    h, w = a.shape
    c = spline_prefilter(a)
    cy, outy = zoom_coords(h, oh)
    cx, outx = zoom_coords(w, ow)
    out = np.zeros((oh, ow), np.float64)
    for oy in range(oh):
        if outy[oy]:
            continue
        sy, wy = _cubic_taps(cy[oy])
        for ox in range(ow):
            if outx[ox]:
                continue
            sx, wx = _cubic_taps(cx[ox])
            acc = 0.0
            for i in range(4):
                row = c[_mirror(sy + i, h)]
                for j in range(4):
                    acc += wy[i] * wx[j] * row[_mirror(sx + j, w)]
            out[oy, ox] = acc
    return out.astype(a.dtype)


def zoom_nearest_restated(a: np.ndarray, oh: int, ow: int) -> np.ndarray:
    h, w = a.shape
    cy, outy = zoom_coords(h, oh)
    cx, outx = zoom_coords(w, ow)
    iy = np.minimum(np.floor(cy + 0.5).astype(np.int64), h - 1)
    ix = np.minimum(np.floor(cx + 0.5).astype(np.int64), w - 1)
    out = a[iy][:, ix].copy()
    out[outy, :] = 0
    out[:, outx] = 0
    return out


# ------------------------------------------------------------------------------------------------ augmentation
def noise_field(seed: int, h: int, w: int, sigma: float) -> np.ndarray:
    """Counter-based normal deviates, one per pixel: two rounds of a 32-bit mix of (seed, pixel index) give two uniforms,
    Box-Muller in float32.  The HIP kernel computes the same integers and the same float32 formula."""
    idx = np.arange(h * w, dtype=np.uint64)

    def mix(v):
        v = v & np.uint64(0xFFFFFFFF)
        v ^= v >> np.uint64(16)
        v = (v * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
        v ^= v >> np.uint64(15)
        v = (v * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
        v ^= v >> np.uint64(16)
        return v

    a = mix(idx * np.uint64(2) + np.uint64(seed) * np.uint64(0x9E3779B9))
    b = mix(idx * np.uint64(2) + np.uint64(1) + np.uint64(seed) * np.uint64(0x9E3779B9) + np.uint64(0x85EBCA6B))
    u1 = (a.astype(np.float32) + np.float32(1.0)) * np.float32(2.0 ** -32)            # (0, 1]
    u2 = b.astype(np.float32) * np.float32(2.0 ** -32)
    r = np.sqrt(np.float32(-2.0) * np.log(u1, dtype=np.float32), dtype=np.float32)
    n = r * np.cos(np.float32(2.0 * math.pi) * u2, dtype=np.float32)
    return (np.float32(sigma) * n).reshape(h, w).astype(np.float32)


def piecewise_grid(h: int, w: int, nb_rows: int = 4, nb_cols: int = 4) -> np.ndarray:
    """Control points of imgaug 0.4.0's PiecewiseAffine (`_get_transformer`): np.linspace(0, h, nb_rows) x np.linspace(0, w, nb_cols), as
    (row, col) pairs in row-major order -- the grid spans [0, h] x [0, w], one step past the last pixel."""
    y, x = np.linspace(0, h, nb_rows), np.linspace(0, w, nb_cols)
    xx, yy = np.meshgrid(x, y)
    return np.dstack([yy.flat, xx.flat])[0]


def piecewise_clip(jitter: np.ndarray, h: int, w: int) -> np.ndarray:
    """Control-point displacements (row, col) after imgaug's restriction of the moved points to the image plane
    (`points_dest = clip(points_src + jitter, 0, size - 1)`): what the augmentation records carry."""
    src = piecewise_grid(h, w)
    dst = src + np.asarray(jitter, np.float64).reshape(-1, 2)
    dst[:, 0] = np.clip(dst[:, 0], 0, h - 1)
    dst[:, 1] = np.clip(dst[:, 1], 0, w - 1)
    return (dst - src).astype(np.float32)


def piecewise_source(disp: np.ndarray, h: int, w: int):
    """Source coordinates (row, col) of every output pixel under imgaug's PiecewiseAffine with control-point displacements `disp`
    ([16, 2] (row, col), already clipped): skimage's `PiecewiseAffineTransform.estimate(src, dst)` -- Delaunay triangulation of the SOURCE
    grid (scipy.spatial, as skimage calls it), one affine map per triangle taking its source corners to its moved corners -- evaluated at
    the output pixel coordinates as `skimage.transform.warp` does (a pixel outside every triangle maps to -1, i.e. to cval; none is:
    the grid spans the slice).  Restated from the published algorithm: skimage cannot be imported here."""
    from scipy import spatial
    src = piecewise_grid(h, w)
    dst = src + np.asarray(disp, np.float64).reshape(-1, 2)
    sxy, dxy = src[:, ::-1], dst[:, ::-1]                             # skimage works in (x, y)
    tri = spatial.Delaunay(sxy)
    cols, rows = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    coords = np.stack([cols.ravel(), rows.ravel()], 1)
    simplex = tri.find_simplex(coords)
    out = np.full_like(coords, -1.0)
    for i, verts in enumerate(tri.simplices):
        mask = simplex == i
        if not mask.any():
            continue
        M = np.linalg.solve(np.hstack([sxy[verts], np.ones((3, 1))]), dxy[verts])        # [x y 1] M = (x', y') at the three corners
        out[mask] = np.hstack([coords[mask], np.ones((int(mask.sum()), 1))]) @ M
    # a map solved from three corners carries ~1e-15 of rounding: on the grid's outer edges (row 0, column 0: the unmoved, clipped corners)
    # that noise alone would decide between the slice's first row and cval.  Coordinates within 1e-9 of an integer are that integer.
    near = np.abs(out - np.round(out)) < 1e-9
    out = np.where(near, np.round(out), out)
    return out[:, 1].reshape(h, w), out[:, 0].reshape(h, w)


def sample_linear(a: np.ndarray, sy: np.ndarray, sx: np.ndarray) -> np.ndarray:
    """scipy order-1 / mode='constant' / cval=0 sampling: a coordinate outside [0, n-1] gives 0."""
    h, w = a.shape
    inside = (sy >= 0) & (sy <= h - 1) & (sx >= 0) & (sx <= w - 1)
    sy, sx = np.where(inside, sy, 0.0), np.where(inside, sx, 0.0)
    y0 = np.minimum(np.floor(sy).astype(np.int64), h - 2) if h > 1 else np.zeros_like(sy, np.int64)
    x0 = np.minimum(np.floor(sx).astype(np.int64), w - 2) if w > 1 else np.zeros_like(sx, np.int64)
    fy, fx = sy - y0, sx - x0
    a64 = a.astype(np.float64)
    v = ((1 - fy) * (1 - fx) * a64[y0, x0] + (1 - fy) * fx * a64[y0, x0 + 1]
         + fy * (1 - fx) * a64[y0 + 1, x0] + fy * fx * a64[y0 + 1, x0 + 1])
    return np.where(inside, v, 0.0)


def sample_nearest(a: np.ndarray, sy: np.ndarray, sx: np.ndarray) -> np.ndarray:
    """scipy order-0 / mode='constant': index floor(c + 0.5); a coordinate outside [0, n-1] gives 0."""
    h, w = a.shape
    inside = (sy >= 0) & (sy <= h - 1) & (sx >= 0) & (sx <= w - 1)
    iy = np.clip(np.floor(sy + 0.5).astype(np.int64), 0, h - 1)
    ix = np.clip(np.floor(sx + 0.5).astype(np.int64), 0, w - 1)
    return np.where(inside, a[iy, ix], 0).astype(a.dtype)


def augment_slice(image: np.ndarray, label: np.ndarray, aug: dict):
    """One slice through the augmentation stage: geometric warp (affine `m` = 2x3 output->source map in (row, col) coordinates,
    plus the optional piecewise-affine warp of a 4x4 control grid, `piecewise_source`), then the pixel stages -- Gaussian blur sigma 1, linear contrast
    center + alpha (v - center), additive Gaussian noise -- in the order `pixel_order` gives (default blur -> contrast -> noise).  `aug` keys: m (6 floats), order (0|1), disp (32 floats or None),
    blur (bool), alpha, center, noise_sigma, noise_seed.  The label is always sampled order 0 and gets no intensity change."""
    if "stages" in aug:                          # a chain of single-augmenter stages, each resampling the previous one's result:
        for st in aug["stages"]:                 # what imgaug's SomeOf(random_order=True) does (dataset_synapse.py:84-95)
            image, label = augment_slice(image, label, st)
        return image.astype(np.float32), label
    h, w = image.shape
    m = np.asarray(aug.get("m", (1, 0, 0, 0, 1, 0)), np.float64)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")

    def source(py, px):
        qy, qx = py, px
        if aug.get("disp") is not None:
            qy, qx = piecewise_source(np.asarray(aug["disp"], np.float32), h, w)
        return (m[2] + m[0] * qy) + m[1] * qx, (m[5] + m[3] * qy) + m[4] * qx      # scipy's order: shift first

    sy, sx = source(yy, xx)
    order = int(aug.get("order", 1))
    img = (sample_linear(image, sy, sx) if order == 1 else sample_nearest(image, sy, sx)).astype(np.float32)
    lab = sample_nearest(label, sy, sx)
    alpha, center = np.float32(aug.get("alpha", 1.0)), np.float32(aug.get("center", 0.0))
    # pixel stages in the drawn order (imgaug SomeOf(random_order=True)); stages that were not drawn follow in the canonical order
    # blur -> contrast -> noise, where they are identities (no blur flag / alpha 1 / sigma 0)
    stages = list(aug.get("pixel_order", ())) + [s_ for s_ in ("blur", "contrast", "noise") if s_ not in aug.get("pixel_order", ())]
    for stage in stages:
        if stage == "blur" and aug.get("blur"):
            img = ndimage.gaussian_filter(img, 1.0, mode="mirror", truncate=float(BLUR_RADIUS)).astype(np.float32)
        elif stage == "contrast":
            img = (center + alpha * (img - center)).astype(np.float32)
        elif stage == "noise" and aug.get("noise_sigma", 0.0) > 0:
            img = (img + noise_field(int(aug["noise_seed"]), h, w, float(aug["noise_sigma"]))).astype(np.float32)
    return img.astype(np.float32), lab


def preprocess_slice(image: np.ndarray, label: np.ndarray, aug: dict, size: int):
    """augment -> resize -> normalise: what one __getitem__ + the trainer's transforms deliver for one slice."""
    img, lab = augment_slice(image, label, aug) if aug is not None else (image, label)
    return resize_normalize(img, lab, size)
