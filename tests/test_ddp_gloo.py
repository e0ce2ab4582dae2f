"""world_size-2 gloo checks of the multi-GPU host logic: the gradient all-reduce over the flat arena
(C1) and the Dice/CE sums all-reduce (C2) that makes the sharded loss equal the single-process global-batch loss."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.transception_oracle import ce_dice_loss, dice_loss_sums
    from transception_amd.train import allreduce_gradients, loss_from_sums, seg_sums_allreduce
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(4, 9, 16, 16, generator=g)
    labels = torch.randint(0, 9, (4, 16, 16), generator=g)
    full, _, _ = ce_dice_loss(logits, labels, 9)
    lo, la = logits[2 * rank:2 * rank + 2], labels[2 * rank:2 * rank + 2]
    prob = torch.softmax(lo, 1)
    inter, ysum, zsum = dice_loss_sums(prob, la, 9)
    ce_sum = torch.nn.functional.cross_entropy(lo, la, reduction="sum")
    sums = torch.zeros(28)
    sums[0] = ce_sum
    sums[1::3], sums[2::3], sums[3::3] = inter, ysum, zsum
    sums, n_pix, w = seg_sums_allreduce(sums, float(lo.shape[0] * 16 * 16))
    loss, ce, dice = loss_from_sums(sums, n_pix, 0.4, 0.6)
    ok_loss = abs(loss.item() - full.item()) < 1e-6 and w == 2

    class Fake:
        _gflat = torch.full((1000,), float(rank + 1))
    allreduce_gradients(Fake)
    ok_grad = bool((Fake._gflat == 3.0).all())

    class Sparse:                                   # live gradients in two far-apart ranges: only those travel
        _gflat = torch.full((2_000_000,), float(rank + 1))
        _used_views = {0: (0, (10, 10)), 1: (128, (50,)), 2: (1_500_000, (1000,))}
    from transception_amd.train import gradient_buckets
    ok_grad = ok_grad and gradient_buckets(Sparse) == [(0, 178), (1_500_000, 1_501_000)]
    allreduce_gradients(Sparse)
    g2 = Sparse._gflat
    ok_grad = ok_grad and bool((g2[:178] == 3.0).all() and (g2[1_500_000:1_501_000] == 3.0).all() and
                               (g2[178:1_500_000] == float(rank + 1)).all())
    # the overlapped form: buckets at/above the encoder boundary travel first (asynchronously), the rest afterwards
    from transception_amd.train import split_buckets
    assert split_buckets([(0, 100), (200, 400), (500, 600)], 300) == ([(0, 100), (200, 300)], [(300, 400), (500, 600)])

    class Two(Sparse):
        _gflat = torch.full((2_000_000,), float(rank + 1))

        @staticmethod
        def late_gradient_offset():
            return 100
    works = allreduce_gradients(Two, None, "late", async_op=True)
    ok_grad = ok_grad and len(works) == 2                      # (100,178) and the far range
    for w in works:
        w.wait()
    g3 = Two._gflat
    ok_grad = ok_grad and bool((g3[:100] == float(rank + 1)).all() and (g3[100:178] == 3.0).all() and (g3[1_500_000:1_501_000] == 3.0).all())
    for w in allreduce_gradients(Two, None, "early", async_op=True):
        w.wait()
    ok_grad = ok_grad and bool((g3[:178] == 3.0).all() and (g3[178:1_500_000] == float(rank + 1)).all())
    # a piece of several small non-adjacent buckets as ONE collective (the exposed last piece of the split backward sweep)

    class Three(Sparse):
        _gflat = torch.arange(2_000_000, dtype=torch.float32) * float(rank + 1)
        _used_views = {0: (0, (10, 10)), 1: (500_000, (64,)), 2: (1_500_000, (1000,))}
    want = torch.arange(2_000_000, dtype=torch.float32) * 3.0
    for async_op in (True, False):
        Three._gflat = torch.arange(2_000_000, dtype=torch.float32) * float(rank + 1)
        works = allreduce_gradients(Three, None, async_op=async_op, ranges=[(0, 600_000), (1_400_000, 2_000_000)], coalesce=True)
        ok_grad = ok_grad and len(works) == (1 if async_op else 0)                 # three buckets, one collective
        for w in works:
            w.wait()
        g4 = Three._gflat
        for a, b in ((0, 100), (500_000, 500_064), (1_500_000, 1_501_000)):
            ok_grad = ok_grad and bool((g4[a:b] == want[a:b]).all())
        ok_grad = ok_grad and bool((g4[100:500_000] == want[100:500_000] / 3.0 * (rank + 1)).all())    # everything else untouched
    # one collective per stop of the split backward sweep (train.comm_schedule / allreduce_scheduled): dead-only gaps are spanned, a
    # piece's stragglers are deferred to the packed last piece, words of a later piece are never touched early
    from transception_amd.train import allreduce_scheduled, comm_schedule

    class Pieces(Sparse):
        _used_views = {0: (0, (1000,)), 1: (10_000, (390_000,)), 2: (400_000, (800_000,)), 3: (1_250_000, (10_000,)),
                       4: (1_600_000, (800_000,)), 5: (2_700_000, (300_000,))}

        @staticmethod
        def gradient_pieces():
            return [("enc", [(1_500_000, 3_000_000)]), ("s3", [(5_000, 1_230_000)]), (None, [(0, 5_000), (1_230_000, 1_500_000)])]
    base = torch.arange(3_000_000, dtype=torch.float32) % 977 + 1.0
    Pieces._gflat = base * float(rank + 1)
    sched = comm_schedule(Pieces)
    ok_grad = ok_grad and [len(e) for _, e in sched] == [1, 1, 1] and [e[0][0] for _, e in sched] == ["span", "span", "pack"]
    ok_grad = ok_grad and sched[0][1][0][1:] == (1_600_000, 3_000_000) and sched[1][1][0][1:] == (5_000, 1_230_000)
    seen = Pieces._gflat.clone()
    for stop, entries in sched:
        for w in allreduce_scheduled(Pieces, entries, None, async_op=True):
            w.wait()
        if stop == "enc":                                      # nothing below the encoder piece has moved yet
            ok_grad = ok_grad and bool((Pieces._gflat[:1_500_000] == seen[:1_500_000]).all())
    for off, shape in Pieces._used_views.values():
        ok_grad = ok_grad and bool((Pieces._gflat[off:off + shape[0]] == 3.0 * base[off:off + shape[0]]).all())
    ret[rank] = (ok_loss, ok_grad)
    dist.destroy_process_group()


def test_two_rank_loss_and_gradient_allreduce():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] == (True, True) for r in range(world)), dict(ret)
