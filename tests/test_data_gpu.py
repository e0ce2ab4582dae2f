"""Parity of the device input pipeline (csrc/data.hip through the C ABI) with the oracle (oracle/data_oracle.py = scipy, the
library the reference calls) and with the fixture generated from the reference's dataset code.

Bars: labels (order-0 sampling, integer work) bit-exact; resized image within 1e-6 of scipy's float32 result before
normalisation (2e-6 after the /0.5); augmented image within 2e-5 (float32 log/cos of the noise generator, sigma 1.275).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import load  # noqa: E402
from oracle import data_oracle as O  # noqa: E402
from transception_amd import data as D  # noqa: E402
from transception_amd._lib import lib  # noqa: E402

DEV = "cuda:0"


def _pair(seed, n, m=None, batch=None):
    g = np.random.default_rng(seed)
    m = m or n
    shape = (n, m) if batch is None else (batch, n, m)
    return g.random(shape).astype(np.float32), g.integers(0, 9, shape).astype(np.float32)


def _dev(img, lab):
    return torch.from_numpy(img).to(DEV), torch.from_numpy(lab.astype(np.uint8)).to(DEV)


@pytest.mark.parametrize("n_in,n_out,batch", [(512, 224, 3), (64, 28, 2), (96, 40, 2), (224, 224, 2), (70, 224, 1)])
def test_resize_normalize_matches_scipy(n_in, n_out, batch):
    img, lab = _pair(11, n_in, batch=batch)
    x, y = D.preprocess_batch(*_dev(img, lab), None, n_out)
    torch.cuda.synchronize()
    assert x.shape == (batch, 1, n_out, n_out) and x.dtype == torch.float32 and y.dtype == torch.int64
    for b in range(batch):
        wx, wy = O.resize_normalize(img[b], lab[b], n_out)
        np.testing.assert_allclose(x[b].cpu().numpy(), wx, atol=2e-6, rtol=0)
        np.testing.assert_array_equal(y[b].cpu().numpy(), wy)
    if n_in == 512:                                                  # the reference's zeroed last row / column
        assert bool((x[:, 0, -1, :] == -1).all()) and bool((x[:, 0, :, -1] == -1).all()) and bool((y[:, -1, :] == 0).all())


def test_spline_coefficients_and_rectangular_zoom():
    """tc_spline_prefilter alone (ragged sizes: rows not a multiple of 64, width 65 = one element into a second chunk) and a
    non-square zoom through the C entry."""
    from scipy import ndimage
    L = lib()
    for (B, H, W) in ((2, 37, 53), (1, 70, 65), (3, 9, 129), (1, 130, 64), (1, 20, 192), (1, 9, 1024), (2, 40, 384), (1, 10, 768), (1, 70, 256), (1, 300, 128)):
        img, lab = _pair(5, H, W, batch=B)
        d_img, d_lab = _dev(img, lab)
        coef = torch.empty((B, H, W), dtype=torch.float64, device=DEV)
        s = torch.cuda.current_stream().cuda_stream
        L.tc_spline_prefilter(d_img.data_ptr(), coef.data_ptr(), B, H, W, s)
        for b in range(B):
            want = ndimage.spline_filter(img[b], 3, output=np.float64, mode="mirror")
            np.testing.assert_allclose(coef[b].cpu().numpy(), want, atol=1e-12, rtol=0)
        OH, OW = max(2, H // 2 + 3), max(2, (W * 2) // 3)
        x = torch.empty((B, 1, OH, OW), dtype=torch.float32, device=DEV)
        y = torch.empty((B, OH, OW), dtype=torch.int64, device=DEV)
        L.tc_zoom_normalize(coef.data_ptr(), d_img.data_ptr(), d_lab.data_ptr(), x.data_ptr(), y.data_ptr(), B, H, W, OH, OW, 0.0, 1.0, s)
        for b in range(B):
            wi = ndimage.zoom(img[b], (OH / H, OW / W), order=3)
            wl = ndimage.zoom(lab[b], (OH / H, OW / W), order=0)
            assert wi.shape == (OH, OW)
            np.testing.assert_allclose(x[b, 0].cpu().numpy(), wi, atol=1e-6, rtol=0)
            np.testing.assert_array_equal(y[b].cpu().numpy(), wl.astype(np.int64))


def test_reference_fixture_through_the_device():
    """The reference's RandomGenerator outputs (rot90/flip, order-0 rotate, zoom) reproduced by the HIP path."""
    G = load("data_pipeline.npz")
    for i in range(int(G["generator/count"][0])):
        seed, n_in, n_out, kind, k, axis, angle = (int(v) for v in G[f"generator/{i}/meta"])
        img, lab = _pair(seed, n_in)
        aug = None
        if kind == 1:
            aug = D.SliceAugmentation(m=D.affine_rot90_flip(k, axis, n_in), order=0)
        elif kind == 2:
            aug = D.SliceAugmentation(m=D.affine_rotate(angle, n_in, n_in), order=0)
        d_img, d_lab = _dev(img[None], lab[None])
        x, y = D.preprocess_batch(d_img, d_lab, [aug], n_out, mean=0.0, std=1.0)
        np.testing.assert_allclose(x[0, 0].cpu().numpy(), G[f"generator/{i}/image"], atol=1e-6, rtol=0)
        np.testing.assert_array_equal(y[0].cpu().numpy(), G[f"generator/{i}/label"].astype(np.int64))
    L = lib()
    for s in range(4):                                               # the two augmentations on their own, exact (order 0)
        seed, n = (int(v) for v in G[f"input/{s}/seed_n"])
        img, lab = _pair(seed, n)
        d_img, d_lab = _dev(img[None], lab[None])
        for tag, aug in (("rot_flip", D.SliceAugmentation(m=D.affine_rot90_flip(*(int(v) for v in G[f"rot_flip/{s}/params"]), n), order=0)),
                         ("rotate", D.SliceAugmentation(m=D.affine_rotate(int(G[f"rotate/{s}/params"][0]), n, n), order=0))):
            rec = torch.from_numpy(D.pack_records([aug])).to(DEV)
            oi, ol = torch.empty_like(d_img), torch.empty_like(d_lab)
            L.tc_slice_augment(d_img.data_ptr(), d_lab.data_ptr(), rec.data_ptr(), oi.data_ptr(), ol.data_ptr(), 1, n, n,
                               torch.cuda.current_stream().cuda_stream)
            np.testing.assert_array_equal(oi[0].cpu().numpy(), G[f"{tag}/{s}/image"])
            np.testing.assert_array_equal(ol[0].cpu().numpy(), G[f"{tag}/{s}/label"])


def _augment_on_device(img, lab, augs, all_rounds=False):
    """The slices' stage chains through tc_slice_augment_chain: the rounds the longest chain needs, or (a captured step) all MAX_ROUNDS."""
    B, H, W = img.shape
    d_img, d_lab = _dev(img, lab)
    rec_np, n = D.pack_rounds(augs)
    rec = torch.from_numpy(rec_np).to(DEV)
    R = D.MAX_ROUNDS
    n = R if all_rounds else max(n, 1)
    bufs = [torch.full_like(d_img, float("nan")) for _ in range(2)] + [torch.full_like(d_lab, 255) for _ in range(2)]
    lib().tc_slice_augment_chain(d_img.data_ptr(), d_lab.data_ptr(), rec.data_ptr(), R - n, R, bufs[0].data_ptr(), bufs[2].data_ptr(),
                                 bufs[1].data_ptr(), bufs[3].data_ptr(), B, H, W, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return bufs[(R - 1) & 1].cpu().numpy(), bufs[2 + ((R - 1) & 1)].cpu().numpy()


def test_augmentation_stage_matches_oracle():
    H, W = 100, 136                                                  # not multiples of the 32x32 tile
    explicit = [
        D.SliceAugmentation(),                                                                     # nothing
        D.SliceAugmentation(blur=True),
        D.SliceAugmentation(alpha=1.37, noise_sigma=D.NOISE_SCALE, noise_seed=12345),
        D.SliceAugmentation(m=D.affine_rotate_xy(33.0, H, W), blur=True, alpha=0.6),
        D.SliceAugmentation(m=D.compose(D.affine_scale(0.7, 1.6, H, W), D.affine_shear(-12.0, H, W))),
        D.SliceAugmentation(m=D.affine_translate(0.15, -0.1, H, W), order=0),
        D.SliceAugmentation(disp=D.piecewise_disp(np.random.default_rng(4).normal(0, 1, (4, 4, 2)) * 3.0, H, W), shape=(H, W)),
        D.SliceAugmentation(disp=D.piecewise_disp(np.random.default_rng(6).normal(0, 1, (4, 4, 2)) * 9.0, H, W), shape=(H, W), order=0),
        D.SliceAugmentation(m=D.affine_flip(0, H, W), disp=D.piecewise_disp(np.random.default_rng(5).normal(0, 1, (4, 4, 2)) * 2.0, H, W),
                            shape=(H, W), blur=True, alpha=1.2, noise_sigma=0.3, noise_seed=2 ** 31 - 2),
    ]
    # every order of the three pixel stages (imgaug applies its augmenters in the drawn order, dataset_synapse.py:84-95): noise drawn
    # before the blur is blurred with the slice, noise drawn before the contrast change is scaled by it
    import itertools
    for perm in itertools.permutations(("blur", "contrast", "noise")):
        explicit.append(D.SliceAugmentation(m=D.affine_rotate_xy(-17.0, H, W), blur=True, alpha=1.45, center=0.4, noise_sigma=0.2,
                                            noise_seed=777, pixel_order=perm))
    explicit.append(D.SliceAugmentation(alpha=0.55, noise_sigma=0.25, noise_seed=5, pixel_order=("noise", "contrast")))
    sampler = D.AugmentSampler(99)
    augs = explicit + [sampler.sample(H, W) for _ in range(24)]
    assert any(a.pixel_order[:1] == ("noise",) and len(a.pixel_order) > 1 for a in augs)
    assert any(a.stages is not None and sum(st.warps() for st in a.stages) >= 2 for a in augs)     # chains with two resamplings
    img, lab = _pair(21, H, W, batch=len(augs))
    oi, ol = _augment_on_device(img, lab, augs)
    for b, a in enumerate(augs):
        wi, wl = O.augment_slice(img[b], lab[b].astype(np.uint8), a.as_dict())
        np.testing.assert_array_equal(ol[b], wl, err_msg=f"label of slice {b} ({a.names})")
        np.testing.assert_allclose(oi[b], wi, atol=2e-5, rtol=0, err_msg=f"image of slice {b} ({a.names})")
    # all MAX_ROUNDS rounds (what a captured step launches whatever the batch holds): the skipped rounds change nothing, bit for bit
    oi4, ol4 = _augment_on_device(img, lab, augs, all_rounds=True)
    np.testing.assert_array_equal(oi4, oi)
    np.testing.assert_array_equal(ol4, ol)
    # a batch without any stage: one copy per slice, in the last round
    oi0, ol0 = _augment_on_device(img[:3], lab[:3], [None, D.SliceAugmentation(stages=[]), None], all_rounds=True)
    np.testing.assert_array_equal(oi0, img[:3])
    np.testing.assert_array_equal(ol0, lab[:3].astype(np.uint8))


def test_loader_batches_match_oracle(tmp_path):
    """npz files -> DeviceLoader -> (x, y) against the oracle pipeline slice by slice, two ranks, real slice size."""
    base, lists = str(tmp_path / "train_npz"), str(tmp_path / "lists")
    D.write_synthetic_synapse(base, lists, n_cases=2, slices_per_case=6, size=512, seed=5)
    ds = D.SynapseSlices(base, lists)
    seen = []
    for rank in range(2):
        loader = D.DeviceLoader(ds, batch_size=3, img_size=224, device=DEV, seed=77, rank=rank, world=2, augment=True, epochs=1)
        assert len(loader) == 2
        n = 0
        for x, y in loader:
            names, augs = loader.last_names, loader.last_augs
            torch.cuda.synchronize()
            xs, ys = x.cpu().numpy(), y.cpu().numpy()
            for j, name in enumerate(names):
                img, lab, _ = ds[ds.sample_list.index(name)]
                wx, wy = O.preprocess_slice(img, lab, augs[j].as_dict(), 224)
                np.testing.assert_array_equal(ys[j], wy, err_msg=f"{name} {augs[j].names}")
                np.testing.assert_allclose(xs[j], wx, atol=1e-4, rtol=0, err_msg=f"{name} {augs[j].names}")
            seen.append((rank, n, tuple(names)))
            n += 1
        assert n == 2
    order = D.epoch_order(len(ds), 0, 77)
    for rank, i, names in seen:
        want = [ds.sample_list[int(k)] for k in order[i * 6 + rank * 3: i * 6 + (rank + 1) * 3]]
        assert list(names) == want


def test_loader_raw_slots_with_consumer_side_preprocessing_match_oracle(tmp_path):
    """DeviceLoader.iter_raw + slot_preprocess (the form a captured step uses: the preprocessing launches belong to the consumer and
    always run MAX_ROUNDS augmentation rounds, identity records in the unused ones): same batches, same results as the oracle, the
    three static slots in rotation, their buffers at the same addresses throughout."""
    base, lists = str(tmp_path / "train_npz"), str(tmp_path / "lists")
    D.write_synthetic_synapse(base, lists, n_cases=2, slices_per_case=6, size=512, seed=5)
    ds = D.SynapseSlices(base, lists)
    loader = D.DeviceLoader(ds, batch_size=3, img_size=224, device=DEV, seed=77, rank=0, world=1, augment=True, epochs=1)
    x = torch.empty((3, 1, 224, 224), dtype=torch.float32, device=DEV)
    y = torch.empty((3, 224, 224), dtype=torch.int64, device=DEV)
    ptrs, n = {}, 0
    for slot in loader.iter_raw(3):
        assert slot["index"] == n % 3
        assert ptrs.setdefault(slot["index"], slot["raw"][0].data_ptr()) == slot["raw"][0].data_ptr()
        names, augs = loader.last_names, loader.last_augs
        loader.slot_preprocess(slot)(x, y)
        torch.cuda.synchronize()
        xs, ys = x.cpu().numpy(), y.cpu().numpy()
        for j, name in enumerate(names):
            img, lab, _ = ds[ds.sample_list.index(name)]
            wx, wy = O.preprocess_slice(img, lab, augs[j].as_dict(), 224)
            np.testing.assert_array_equal(ys[j], wy, err_msg=f"{name} {augs[j].names}")
            np.testing.assert_allclose(xs[j], wx, atol=1e-4, rtol=0, err_msg=f"{name} {augs[j].names}")
        n += 1
    assert n == 4 and not loader._thread.is_alive()


def test_loader_without_augmentation_and_early_exit(tmp_path):
    base, lists = str(tmp_path / "train_npz"), str(tmp_path / "lists")
    D.write_synthetic_synapse(base, lists, n_cases=1, slices_per_case=8, size=128, seed=6)
    ds = D.SynapseSlices(base, lists)
    loader = D.DeviceLoader(ds, batch_size=2, img_size=64, device=DEV, augment=False, shuffle=False, epochs=2)
    got = 0
    for x, y in loader:
        names = loader.last_names
        for j, name in enumerate(names):
            img, lab, _ = ds[ds.sample_list.index(name)]
            wx, wy = O.resize_normalize(img, lab, 64)
            np.testing.assert_allclose(x[j].cpu().numpy(), wx, atol=2e-6, rtol=0)
            np.testing.assert_array_equal(y[j].cpu().numpy(), wy)
        got += 1
        if got == 5:
            break                                                    # leaves the producer thread mid-epoch: must shut down cleanly
    assert got == 5 and not loader._thread.is_alive()


def test_c_abi_argument_checks():
    from transception_amd._lib import TcError
    L = lib()
    t = torch.zeros(16, device=DEV)
    with pytest.raises(TcError):
        L.tc_slice_augment(t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), 1, 4, 4, None)      # too small / in place
    with pytest.raises(TcError):
        L.tc_zoom_normalize(None, t.data_ptr(), None, t.data_ptr(), None, 1, 4, 4, 2, 2, 0.5, 0.5, None)              # no coef but resize
    with pytest.raises(TcError):
        L.tc_spline_prefilter(None, t.data_ptr(), 1, 4, 4, None)
