"""Host-side logic that needs no GPU: state_dict schema, constructor contract, no-CPU-fallback, the engine's
gradient-region bookkeeping, the fused-SGD segment table and the cosine schedule."""
import math

import pytest
import torch

from transception_amd import MSTransception, TransCeption
from transception_amd.seeded_init import load_manifest, seeded_state_dict


@pytest.fixture(scope="module")
def model():
    return MSTransception(num_classes=9)


def test_state_dict_schema_matches_reference_manifest(model):
    ents = load_manifest()
    sd = model.state_dict()
    assert list(sd.keys()) == [e[0] for e in ents]
    for k, shape, _ in ents:
        assert tuple(sd[k].shape) == tuple(shape), k
    assert sum(p.numel() for p in model.parameters()) == 47_316_553
    # aliases of the shared cpe / crpe modules share storage
    a = "backbone.mhca_stage2.mhca_blks.0.cpe.proj.weight"
    b = "backbone.mhca_stage2.mhca_blks.0.MHCA_layers.2.cpe.proj.weight"
    assert sd[a].data_ptr() == sd[b].data_ptr()


def test_load_seeded_state_strict(model):
    res = model.load_state_dict(seeded_state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert TransCeption is MSTransception


def test_constructor_contract():
    MSTransception(num_classes=2, head_count=8, dil_conv=1, token_mlp_mode="mix_skip", MSViT_config=2, concat="coord",
                   have_bridge="original", use_sa_config=1, sa_ker=7, Stage_3or4=3, inter="res", num_sp=1,
                   br_ch_att_list=[True, False, False, False])
    for kw in (dict(concat="nonsense"), dict(have_bridge="sp", num_sp=-1), dict(Stage_3or4=4, concat="3d"), dict(token_mlp_mode="mlp")):
        with pytest.raises(NotImplementedError):
            MSTransception(**kw)


def test_no_cpu_fallback(model):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(torch.zeros(1, 1, 224, 224))


def test_engine_gradient_regions():
    from transception_amd.engine import Graph, Var
    G = Graph(torch.float32, torch.device("cpu"), training=True, record=True)
    root = Var(torch.zeros(10, 12))
    q, v = root.colslice(0, 4), root.colslice(8, 12)
    g, acc = G.wgrad(v.colslice(0, 2))
    assert acc == 0 and tuple(g.shape) == (10, 2)
    g, acc = G.wgrad(v)                       # overlaps the sub-slice written above -> must accumulate
    assert acc == 1
    assert G.wgrad(q)[1] == 0 and G.wgrad(q)[1] == 1
    assert G.grad_of(root.colslice(4, 8)) is None          # untouched columns
    assert G.wgrad(root)[1] == 1                           # whole write after parts accumulates
    r2 = Var(torch.zeros(6, 4))
    assert G.wgrad(r2)[1] == 0 and G.wgrad(r2.rowslice(0, 3))[1] == 1
    r3 = Var(torch.zeros(8, 4))
    blk = r3.rowslice(2, 6).reshape(2, 8)
    assert blk.region == (2, 6, 0, 4) and not blk.is_whole
    assert G.wgrad(blk)[1] == 0 and G.wgrad(r3.rowslice(4, 8))[1] == 1 and G.wgrad(r3.rowslice(0, 2))[1] == 0
    src = torch.ones(8, 4)
    r4 = Var(torch.zeros(8, 4))
    G.pass_grad(r4, src)                                    # alias, no copy
    assert G.grad_of(r4).data_ptr() == src.data_ptr()


def test_cosine_schedule_matches_torch():
    from transception_amd.train import cosine_lr
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=0.05)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=100)
    for step in range(1, 6):
        opt.step()
        sch.step()
        assert math.isclose(opt.param_groups[0]["lr"], cosine_lr(0.05, step, 100), rel_tol=1e-9)


def test_eval_dice_conventions():
    """dice_from_counts (product) == eval_dice (oracle) == the reference's calculate_metric_percase conventions (utils.py:50-60)."""
    import numpy as np
    import torch
    from oracle.transception_oracle import eval_argmax_counts, eval_dice
    from transception_amd.evaluate import dice_from_counts
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(2, 5, 8, 8, generator=g)
    labels = torch.randint(0, 4, (2, 8, 8), generator=g)           # class 4 never appears in the labels
    logits[:, 3] = -50.0                                            # class 3 is never predicted
    pred, counts = eval_argmax_counts(logits, labels, 5)
    d_oracle = eval_dice(counts)
    d_prod = dice_from_counts(counts.numpy())
    assert np.allclose(d_oracle, d_prod)
    p1, g1 = (pred == 1), (labels == 1)
    assert abs(d_prod[0] - 2.0 * (p1 & g1).sum().item() / (p1.sum().item() + g1.sum().item())) < 1e-12
    assert d_prod[2] == 0.0                                          # never predicted -> 0
    assert d_prod[3] in (0.0, 1.0)                                   # label-free class: 1 if predicted anywhere, else 0


def test_hd95_follows_its_definition():
    """transception_amd.evaluate.hd95 (scipy restatement of medpy's algorithm, utils.py:55) against the brute-force oracle."""
    import numpy as np
    from oracle.transception_oracle import eval_hd95
    from transception_amd.evaluate import calculate_metric_percase, hd95
    g = np.random.default_rng(0)
    yy, xx = np.mgrid[0:40, 0:48]
    for t in range(6):
        cy, cx, r = g.uniform(12, 28), g.uniform(12, 36), g.uniform(4, 10)
        a = (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
        b = (yy - cy - g.uniform(-4, 4)) ** 2 / 1.5 + (xx - cx - g.uniform(-4, 4)) ** 2 < (r + g.uniform(-2, 2)) ** 2
        if not b.any():
            continue
        assert abs(hd95(a, b) - eval_hd95(a, b)) < 1e-9
        assert abs(hd95(a, b, voxelspacing=(1.0, 2.5)) - eval_hd95(a, b, (1.0, 2.5))) < 1e-9
    zz, yy, xx = np.mgrid[0:12, 0:20, 0:20]
    a = (zz - 6) ** 2 + (yy - 9) ** 2 + (xx - 10) ** 2 < 25
    b = (zz - 5) ** 2 + (yy - 11) ** 2 + (xx - 9) ** 2 < 30
    assert abs(hd95(a, b) - eval_hd95(a, b)) < 1e-9
    assert hd95(a, a) == 0.0
    # calculate_metric_percase conventions (utils.py:50-60)
    d, h = calculate_metric_percase(a.astype(np.uint8), b.astype(np.uint8))
    assert abs(d - 2.0 * (a & b).sum() / (a.sum() + b.sum())) < 1e-12 and abs(h - eval_hd95(a, b)) < 1e-9
    assert calculate_metric_percase(a.astype(np.uint8), np.zeros_like(a, np.uint8)) == (1.0, 0.0)
    assert calculate_metric_percase(np.zeros_like(a, np.uint8), b.astype(np.uint8)) == (0.0, 0.0)


def test_trainer_schedule_helpers():
    from transception_amd.trainer import checkpoint_epochs, scaled_base_lr
    assert checkpoint_epochs(400, 20) == [219, 239, 259, 279, 299, 319, 339, 359, 379, 399]
    assert checkpoint_epochs(3, 20) == [2] and checkpoint_epochs(150, 50)[-1] == 149 and 99 in checkpoint_epochs(150, 50)
    assert scaled_base_lr(0.05, 24) == 0.05 and scaled_base_lr(0.05, 16) == 0.05
    assert abs(scaled_base_lr(0.05, 20) - 0.05 * 20 / 24) < 1e-12


@pytest.mark.parametrize("legs", ["merged", "split"])
def test_gradient_pieces_partition_the_arena_and_follow_the_backward_order(model, legs, monkeypatch):
    """model.gradient_pieces(): what a multi-GPU step sends after each leg of its backward sweep (bridge + decoders, stages 4 + 3 -- one
    piece by default, one each with TC_GRAD_LEGS=split -- the rest) -- disjoint ranges that cover the whole flat gradient arena, each
    made of whole modules."""
    import math
    import torch
    from transception_amd.train import clip_buckets
    monkeypatch.setenv("TC_GRAD_LEGS", legs)
    model._ensure_flat(torch.device("cpu"))
    pieces = model.gradient_pieces()
    assert [u for u, _ in pieces] == (["encoder_done", "stage4_done", "stage3_done", None] if legs == "split" else ["encoder_done", "stage3_done", None])
    flat = sorted(r for _, rs in pieces for r in rs)
    assert flat[0][0] == 0 and flat[-1][1] == model._gflat.numel()
    assert all(a[1] == b[0] for a, b in zip(flat, flat[1:]))                 # no hole, no overlap
    owner = {}
    for until, rs in pieces:
        for lo, hi in rs:
            for name, (off, shape) in model._index.items():
                if lo <= off < hi:
                    assert off + math.prod(shape) <= hi                      # no tensor straddles a cut
                    owner[name] = until
    assert all(owner[n] == ("stage4_done" if legs == "split" else "stage3_done") for n in owner
               if n.startswith(("backbone.mhca_stage4.", "backbone.patch_embed_stage4.")))
    assert all(owner[n] == "stage3_done" for n in owner if n.startswith(("backbone.mhca_stage3.", "backbone.patch_embed_stage3.")))
    assert all(owner[n] == "encoder_done" for n in owner if not n.startswith("backbone."))
    assert all(owner[n] is None for n in owner if n.startswith(("backbone.block1.", "backbone.patch_embed1.", "backbone.mhca_stage2.")))
    s4 = sum(hi - lo for lo, hi in pieces[1][1]) / model.late_gradient_offset()
    assert (0.6 < s4 < 0.7) if legs == "split" else (0.85 < s4 < 0.95)        # stage 4 is ~64 % of the encoder's arena, stages 4 + 3 ~90 %
    assert clip_buckets([(0, 100), (200, 400), (500, 600)], [(50, 250), (550, 900)]) == [(50, 100), (200, 250), (550, 600)]


@pytest.mark.parametrize("h,w", [(512, 512), (100, 136), (224, 224)])
def test_piecewise_affine_triangles_follow_the_delaunay_statement(h, w):
    """PiecewiseAffine (dataset_synapse.py:93): the record the HIP kernel reads -- clipped control-point displacements + one diagonal bit per
    cell of imgaug's 4x4 grid (transception_amd.data) and the barycentric mix inside the pixel's triangle (csrc/data.hip: restated here in
    numpy) -- gives the source coordinates of oracle.data_oracle.piecewise_source, which builds skimage's PiecewiseAffineTransform from the
    published algorithm (Delaunay triangulation of the source grid, one affine map per triangle)."""
    import numpy as np
    from oracle import data_oracle as O
    from transception_amd import data as D
    g = np.random.default_rng(h + w)
    jitter = g.normal(0.0, 1.0, (4, 4, 2)) * np.array([0.03 * h, 0.03 * w])
    disp = D.piecewise_disp(jitter, h, w)
    np.testing.assert_array_equal(disp.reshape(-1, 2), O.piecewise_clip(jitter, h, w))
    np.testing.assert_array_equal(D.piecewise_grid(h, w), O.piecewise_grid(h, w))
    bits = D.piecewise_diagonals(h, w)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    gy, gx = yy * (3.0 / h), xx * (3.0 / w)
    y0, x0 = np.minimum(np.floor(gy).astype(int), 2), np.minimum(np.floor(gx).astype(int), 2)
    fy, fx = gy - y0, gx - x0
    d = disp.astype(np.float64)
    tl, tr, bl, br = d[y0, x0], d[y0, x0 + 1], d[y0 + 1, x0], d[y0 + 1, x0 + 1]
    diag_b = ((bits >> (y0 * 3 + x0)) & 1).astype(bool)
    z = np.zeros_like(fx)
    upper, left = fx >= fy, fx + fy <= 1.0
    l00 = np.where(diag_b, np.where(left, 1 - fx - fy, z), np.where(upper, 1 - fx, 1 - fy))
    l01 = np.where(diag_b, np.where(left, fx, 1 - fy), np.where(upper, fx - fy, z))
    l10 = np.where(diag_b, np.where(left, fy, 1 - fx), np.where(upper, z, fy - fx))
    l11 = np.where(diag_b, np.where(left, z, fx + fy - 1), np.where(upper, fy, fx))
    mix = l00[..., None] * tl + l01[..., None] * tr + l10[..., None] * bl + l11[..., None] * br
    sy, sx = O.piecewise_source(disp, h, w)
    np.testing.assert_allclose(yy + mix[..., 0], sy, rtol=0, atol=1e-9)
    np.testing.assert_allclose(xx + mix[..., 1], sx, rtol=0, atol=1e-9)
    # nothing drawn: the far edge of the grid lies one step outside the slice and is clipped onto its last row / column, nothing else moves
    sy0, sx0 = O.piecewise_source(D.piecewise_disp(np.zeros((4, 4, 2)), h, w), h, w)
    assert np.abs(sy0 - yy)[: h // 3].max() < 1e-9 and np.abs(sx0 - xx)[:, : w // 3].max() < 1e-9
    assert -1.0 <= (sy0 - yy).min() and (sy0 - yy).max() <= 1e-9


def test_comm_schedule_empty_last_piece_and_capped_dead_gaps():
    """ADVICE r5 (train.comm_schedule): (1) a last piece with no live run and nothing deferred yields NO collective (it used to emit an empty
    pack, and torch.cat([]) raised); (2) a dead gap above SPAN_GAP_MAX elements is not sent along to save a collective -- the stop's two
    live runs travel separately; a gap below it still is."""
    import torch
    from transception_amd import train

    class M1:                                       # every live word sits in the first piece
        _gflat = torch.zeros(1_000_000)
        _used_views = {0: (600_000, (100_000,)), 1: (800_000, (50_000,))}

        @staticmethod
        def gradient_pieces():
            return [("enc", [(500_000, 1_000_000)]), (None, [(0, 500_000)])]
    sched = train.comm_schedule(M1)
    assert [stop for stop, _ in sched] == ["enc", None]
    assert sched[1][1] == [] and train.allreduce_scheduled(M1, sched[1][1], None) == []
    assert sum(b - a for e in sched[0][1] for a, b in ([(e[1], e[2])] if e[0] == "span" else e[1])) >= 150_000

    big = train.SPAN_GAP_MAX + 500_000

    class M2:                                       # two live runs in one piece, a dead gap larger than SPAN_GAP_MAX between them
        _gflat = torch.zeros(3_000_000 + big)
        _used_views = {0: (0, (1_000_000,)), 1: (1_000_000 + big, (2_000_000,))}

        @staticmethod
        def gradient_pieces():
            return [("enc", [(0, 3_000_000 + big)]), (None, [])]
    spans = [e for _, ent in train.comm_schedule(M2) for e in ent if e[0] == "span"]
    assert sorted((e[1], e[2]) for e in spans) == [(0, 1_000_000), (1_000_000 + big, 3_000_000 + big)]

    class M3(M2):                                   # the same with a gap just below the limit: one collective, the zeros ride along
        _gflat = torch.zeros(3_000_000 + train.SPAN_GAP_MAX - 8)
        _used_views = {0: (0, (1_000_000,)), 1: (1_000_000 + train.SPAN_GAP_MAX - 8, (2_000_000,))}

        @staticmethod
        def gradient_pieces():
            return [("enc", [(0, 3_000_000 + train.SPAN_GAP_MAX - 8)]), (None, [])]
    spans = [e for _, ent in train.comm_schedule(M3) for e in ent if e[0] == "span"]
    assert [(e[1], e[2]) for e in spans] == [(0, 3_000_000 + train.SPAN_GAP_MAX - 8)]
