"""CPU checks of the input-pipeline row: the oracle against scipy and against the fixture generated from the
reference's own dataset code (tests/golden/make_data_golden.py), and the host logic of transception_amd.data."""
import os

import numpy as np
import pytest
from scipy import ndimage

from golden_util import load
from oracle import data_oracle as O
from transception_amd import data as D


def _pair(seed, n, m=None):
    g = np.random.default_rng(seed)
    m = m or n
    return g.random((n, m)).astype(np.float32), g.integers(0, 9, (n, m)).astype(np.float32)


@pytest.mark.parametrize("shape,out", [((37, 53), (16, 23)), ((64, 64), (28, 28)), ((20, 31), (45, 50)), ((9, 9), (4, 4))])
def test_zoom_restatement_matches_scipy(shape, out):
    img, lab = _pair(1, *shape)
    zf = (out[0] / shape[0], out[1] / shape[1])
    want = ndimage.zoom(img, zf, order=3)
    assert want.shape == out
    got = O.zoom_cubic_restated(img, *out)
    np.testing.assert_allclose(got, want, atol=2e-7, rtol=0)
    np.testing.assert_array_equal(O.zoom_nearest_restated(lab, *out), ndimage.zoom(lab, zf, order=0))


def test_zoom_last_row_quirk_is_real():
    """At 64 -> 28 (and 512 -> 224) the last output coordinate rounds above n-1 and scipy returns cval: the reference trains on
    slices whose last row and column are 0 (-1 after normalisation)."""
    img, lab = _pair(2, 64)
    z = ndimage.zoom(img + 1.0, (28 / 64, 28 / 64), order=3)
    assert np.all(z[-1, :] == 0) and np.all(z[:, -1] == 0) and np.all(z[:-1, :-1] != 0)
    _, out512 = O.zoom_coords(512, 224)
    assert out512.sum() == 1 and out512[-1]
    _, out96 = O.zoom_coords(96, 40)
    assert out96.sum() == 0


def test_prefilter_matches_scipy():
    img, _ = _pair(3, 33, 70)
    np.testing.assert_allclose(O.spline_prefilter(img), ndimage.spline_filter(img, 3, output=np.float64, mode="mirror"), atol=1e-13)


def test_oracle_matches_reference_fixture():
    G = load("data_pipeline.npz")
    for s in range(4):
        seed, n = (int(v) for v in G[f"input/{s}/seed_n"])
        img, lab = _pair(seed, n)
        k, axis = (int(v) for v in G[f"rot_flip/{s}/params"])
        oi, ol = O.augment_slice(img, lab.astype(np.uint8), {"m": D.affine_rot90_flip(k, axis, n), "order": 0})
        np.testing.assert_array_equal(oi, G[f"rot_flip/{s}/image"])
        np.testing.assert_array_equal(ol, G[f"rot_flip/{s}/label"])
        angle = int(G[f"rotate/{s}/params"][0])
        oi, ol = O.augment_slice(img, lab.astype(np.uint8), {"m": D.affine_rotate(angle, n, n), "order": 0})
        np.testing.assert_array_equal(oi, G[f"rotate/{s}/image"])
        np.testing.assert_array_equal(ol, G[f"rotate/{s}/label"])
    for i in range(int(G["generator/count"][0])):
        seed, n_in, n_out, kind, k, axis, angle = (int(v) for v in G[f"generator/{i}/meta"])
        img, lab = _pair(seed, n_in)
        aug = None if kind == 0 else {"m": D.affine_rot90_flip(k, axis, n_in) if kind == 1 else D.affine_rotate(angle, n_in, n_in), "order": 0}
        x, y = O.preprocess_slice(img, lab.astype(np.uint8), aug, n_out)
        np.testing.assert_allclose(x[0] * 0.5 + 0.5, G[f"generator/{i}/image"], atol=1e-6, rtol=0)
        np.testing.assert_array_equal(y, G[f"generator/{i}/label"])
        if n_in <= 64:                                               # the loop restatement too, where it is quick
            ai, al = (img, lab) if aug is None else O.augment_slice(img, lab.astype(np.uint8), aug)
            np.testing.assert_allclose(O.zoom_cubic_restated(ai, n_out, n_out), G[f"generator/{i}/image"], atol=1e-6, rtol=0)
            np.testing.assert_array_equal(O.zoom_nearest_restated(np.asarray(al), n_out, n_out), G[f"generator/{i}/label"])


def test_affine_maps_follow_numpy_and_scipy():
    img, lab = _pair(5, 24)
    for k in range(4):
        for axis in (0, 1):
            want = np.flip(np.rot90(img, k), axis=axis)
            got, _ = O.augment_slice(img, lab.astype(np.uint8), {"m": D.affine_rot90_flip(k, axis, 24), "order": 0})
            np.testing.assert_array_equal(got, want)
    for angle in (-19, 7, 90):
        want = ndimage.rotate(img, angle, order=1, reshape=False)
        got, _ = O.augment_slice(img, lab.astype(np.uint8), {"m": D.affine_rotate(angle, 24, 24), "order": 1})
        np.testing.assert_allclose(got, want, atol=1e-6)
    # compose(first, then): the slice after `first` goes through `then`
    a, b = D.affine_rotate(11, 24, 24), D.affine_flip(1, 24, 24)
    step1, l1 = O.augment_slice(img, lab.astype(np.uint8), {"m": a, "order": 0})
    step2, _ = O.augment_slice(step1, l1, {"m": b, "order": 0})
    both, _ = O.augment_slice(img, lab.astype(np.uint8), {"m": D.compose(a, b), "order": 0})
    np.testing.assert_array_equal(both, step2)
    # the imgaug-style maps: identity parameters give the identity, scale 2 magnifies about the centre
    for m in (D.affine_scale(1, 1, 24, 30), D.affine_rotate_xy(0, 24, 30), D.affine_shear(0, 24, 30), D.affine_translate(0, 0, 24, 30)):
        np.testing.assert_allclose(m, D.IDENTITY, atol=1e-12)
    m = D.affine_scale(2.0, 2.0, 25, 25)
    assert m[0] == pytest.approx(0.5) and m[4] == pytest.approx(0.5) and m[2] == pytest.approx(6.0)
    m = D.affine_translate(0.25, 0.0, 20, 40)                        # content moves right by 10 columns: source column = x - 10
    assert m[5] == pytest.approx(-10.0) and m[2] == pytest.approx(0.0)


def test_sampler_is_seeded_and_in_range():
    a = [D.AugmentSampler(7).sample(512, 512) for _ in range(1)]
    s1, s2 = D.AugmentSampler(7), D.AugmentSampler(7)
    seen = set()
    for _ in range(300):
        x, y = s1.sample(512, 512), s2.sample(512, 512)
        assert x.names == y.names and len(x.stages) == len(y.stages) <= len(x.names) <= 4 and len(set(x.names)) == len(x.names)
        for sx, sy in zip(x.stages, y.stages):                       # one stage per drawn augmenter, in the drawn order
            assert sx.m == sy.m and sx.alpha == sy.alpha and sx.noise_seed == sy.noise_seed and sx.stages is None
            assert 0.5 <= sx.alpha <= 1.5 and sx.noise_sigma in (0.0, D.NOISE_SCALE)
            assert sum([sx.warps(), sx.blur, sx.alpha != 1.0, sx.noise_sigma > 0]) == 1      # exactly one operation per stage
            r = sx.record()
            assert bool(r.flags & D.TC_AUG_WARP) == sx.warps() and bool(r.flags & D.TC_AUG_BLUR) == sx.blur
        seen.update(x.names)
    assert seen == set(D.AugmentSampler.NAMES)
    rec, n = D.pack_rounds([a[0], None])
    assert rec.shape == (D.MAX_ROUNDS, 2, 200) and n == len(a[0].stages)
    # chains end in the last round: skip records before a chain starts, its first stage reads the raw slice, a slice without stages is
    # copied once in the last round
    import ctypes
    from transception_amd._lib import TcSliceAug
    three = D.SliceAugmentation(stages=[D.SliceAugmentation(blur=True), D.SliceAugmentation(alpha=1.2), D.SliceAugmentation(m=D.affine_flip(0, 64, 64))])
    rec, n = D.pack_rounds([three, None, D.SliceAugmentation(blur=True)])
    assert n == 3
    flags = [[TcSliceAug.from_buffer_copy(rec[r, b].tobytes()).flags for b in range(3)] for r in range(D.MAX_ROUNDS)]
    S, F = D.TC_AUG_SKIP, D.TC_AUG_FROM_RAW
    assert [f[0] & (S | F) for f in flags] == [S, F, 0, 0] and flags[1][0] & D.TC_AUG_BLUR and flags[3][0] & D.TC_AUG_WARP
    assert [f[1] & (S | F | D.TC_AUG_WARP | D.TC_AUG_BLUR) for f in flags] == [S, S, S, F]
    assert [f[2] & (S | F) for f in flags] == [S, S, S, F] and flags[3][2] & D.TC_AUG_BLUR
    assert D.pack_records([D.SliceAugmentation(blur=True), None]).shape == (2, 200)


def test_synthetic_set_round_trip_and_sharding(tmp_path):
    names = D.write_synthetic_synapse(str(tmp_path / "train_npz"), str(tmp_path / "lists"), n_cases=2, slices_per_case=5, size=64, seed=3)
    ds = D.SynapseSlices(str(tmp_path / "train_npz"), str(tmp_path / "lists"))
    assert len(ds) == 10 and ds.sample_list == names
    img, lab, name = ds[3]
    raw = np.load(os.path.join(str(tmp_path / "train_npz"), name + ".npz"))
    assert raw["image"].dtype == np.float32 and raw["label"].dtype == np.float32 and raw["image"].shape == (64, 64)
    assert img.dtype == np.float32 and lab.dtype == np.uint8 and 0 <= img.min() and img.max() <= 1 and lab.max() <= 8 and lab.max() > 0
    with pytest.raises(NotImplementedError):
        D.SynapseSlices(str(tmp_path / "train_npz"), str(tmp_path / "lists"), split="test_vol")
    order = D.epoch_order(10, 0, 1234)
    assert sorted(order.tolist()) == list(range(10)) and not np.array_equal(order, D.epoch_order(10, 1, 1234))
    np.testing.assert_array_equal(order, D.epoch_order(10, 0, 1234))
    parts = [D.rank_batches(order, 2, r, 2) for r in range(2)]
    assert len(parts[0]) == len(parts[1]) == 3                      # ceil(10 / (2*2)) global batches, like DataLoader(drop_last=False)
    for i in range(2):
        merged = np.concatenate([parts[0][i], parts[1][i]])
        np.testing.assert_array_equal(merged, order[i * 4:(i + 1) * 4])
    last = np.concatenate([parts[0][2], parts[1][2]])               # the short tail is completed from the head of the same permutation
    np.testing.assert_array_equal(last, np.concatenate([order[8:], order[:2]]))
    assert len(D.rank_batches(np.arange(2211), 24, 0, 1)) == 93     # Synapse at the reference's batch size (trainer.py:104)


def test_direct_npz_reads_match_np_load(tmp_path):
    """read_slice_into (stored members read straight into caller buffers) against np.load, incl. the compressed fallback."""
    base, lists = str(tmp_path / "a"), str(tmp_path / "l")
    D.write_synthetic_synapse(base, lists, n_cases=1, slices_per_case=3, size=48, seed=4)
    ds = D.SynapseSlices(base, lists)
    img, lab = np.empty((48, 48), np.float32), np.empty((48, 48), np.uint8)
    for i in range(3):
        a, b, name = ds[i]
        if i == 2:
            np.savez_compressed(os.path.join(base, name + ".npz"), image=a, label=b.astype(np.float32))
        img[:] = -1
        lab[:] = 255
        assert ds.read_into(i, img, lab) == name
        np.testing.assert_array_equal(img, a)
        np.testing.assert_array_equal(lab, b)
    with pytest.raises(Exception):
        ds.read_into(0, np.empty((32, 32), np.float32), np.empty((32, 32), np.uint8))
