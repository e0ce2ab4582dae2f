"""RCCL executed on hardware before the driver's 8-GPU run: ONE rank on the MI355X, backend "nccl" (= RCCL on ROCm).

A 1-rank RCCL all-reduce still goes through communicator creation, the collective stream, asynchronous work handles between graph
replays and the no-collective-in-capture rule, i.e. everything of the multi-GPU step (trainer.py:110-111's nn.DataParallel replaced by
batch sharding + gradient all-reduce) except the wire.  The captured split step -- forward graph, loss-sum
all-reduce, backward cut at the encoder mark with the bridge/decoder buckets reduced asynchronously, encoder backward, SGD graph -- must
follow the eager step with one synchronous gradient all-reduce."""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    import transception_amd.train as T
    T.COMM_AT_WORLD_1 = True
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from transception_amd import MSTransception
        from transception_amd.seeded_init import seeded_input, seeded_labels, seeded_state_dict
        assert dist.get_backend() == "nccl" and T.comm_on(dist.group.WORLD)
        x = torch.from_numpy(seeded_input(2)).to(dev)
        y = torch.from_numpy(seeded_labels(2)).to(dev)

        def fresh():
            m = MSTransception(num_classes=9)
            m.load_state_dict(seeded_state_dict(), strict=True)
            m.to(dev).train()
            m.set_compute_dtype(torch.float32)
            return m
        me, mg = fresh(), fresh()
        oe, og = T.FusedSGD(me, lr=0.05), T.FusedSGD(mg, lr=0.05)
        le, lg = T.SegLoss(9, group=dist.group.WORLD), T.SegLoss(9, group=dist.group.WORLD)
        g0 = torch.ones(1024, device=dev)
        dist.all_reduce(g0)                                        # the communicator exists and sums one rank
        assert float(g0.sum()) == 1024.0
        for _ in range(2):                                         # = the capture's two warm-up steps
            T.train_step(me, le, oe, x, y, dist.group.WORLD)
        step = T.GraphedStep(mg, lg, og, x, y, dist.group.WORLD, warmup=2, force_split=True)
        assert step.split and step.distributed
        early_b, late_b = T.split_buckets(T.gradient_buckets(mg), mg.late_gradient_offset())
        assert early_b and late_b
        diffs = []
        for _ in range(5):
            a = float(T.train_step(me, le, oe, x, y, dist.group.WORLD)[0])
            b = float(step()[0])
            diffs.append(abs(a - b))
        torch.cuda.synchronize()
        ret[0] = (max(diffs), float((me.flat_parameters() - mg.flat_parameters()).abs().max()), step.kernel_nodes())
    finally:
        dist.destroy_process_group()


def test_one_rank_rccl_split_step_follows_eager():
    ret = mp.Manager().dict()
    port = 29950 + (os.getpid() % 40)
    mp.spawn(_worker, args=(port, ret), nprocs=1, join=True)
    dl, dp, nodes = ret[0]
    assert dl < 2e-5, dict(ret)                              # same losses step by step (fp32 compute)
    assert dp < 1e-4, dict(ret)                              # and the same parameters after 7 steps
    assert nodes is None or nodes > 100                      # the captured graphs hold the step's kernels (counted when the runtime allows)


def test_bench_force_split_runs_on_a_one_rank_rccl_group():
    """`python bench.py --gpus 1 --force-split`: the split step with its collectives on a 1-rank RCCL communicator; the line says so."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-split", "--steps", "3", "--warmup", "1",
                        "--batch", "2", "--no-cpu", "--no-side"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1 and out["config"]["collective_backend"].startswith("nccl")
    assert out["config"]["allreduce_exposed_ms"] is not None and out["value"] > 0
    assert out["config"]["launches_per_step"] is None or out["config"]["launches_per_step"] > 100


def test_comm_plan_covers_every_live_gradient_once():
    """train.comm_plan (the dry description of what an N-GPU step sends): its pieces partition the live part of the gradient arena --
    every bucket element of gradient_buckets() appears in exactly one piece -- and the expected ring times follow from the bytes."""
    from transception_amd import MSTransception
    from transception_amd.seeded_init import seeded_input, seeded_labels, seeded_state_dict
    from transception_amd.train import SegLoss, comm_plan, gradient_buckets
    m = MSTransception(num_classes=9)
    m.load_state_dict(seeded_state_dict(), strict=True)
    m.to("cuda:0").train()
    loss, _, _ = SegLoss(9)(m(torch.from_numpy(seeded_input(1)).to("cuda:0")), torch.from_numpy(seeded_labels(1)).to("cuda:0"))
    loss.backward()
    plan = comm_plan(m, 8)
    lb = gradient_buckets(m)
    live = sum(b - a for a, b in lb)
    sent = sorted((a, b) for p in plan["pieces"] for a, b in p["buckets_elements"])
    # the sent ranges are disjoint, cover every live word exactly once, and whatever else they span is grad-less (zeros): dead words of
    # gaps the schedule sends along instead of opening another collective (train.comm_schedule)
    assert all(sent[i][1] <= sent[i + 1][0] for i in range(len(sent) - 1))
    covered = sum(max(0, min(b, y) - max(a, x)) for a, b in sent for x, y in lb)
    assert covered == live and abs(plan["live_gradient_megabytes_per_step"] - 4 * live / 1e6) < 1e-9
    g = m.flat_gradients()
    for a, b in sent:
        for x, y in [(a, b)]:
            dead = torch.ones(b - a, dtype=torch.bool, device=g.device)
            for u, v in lb:
                lo, hi = max(a, u), min(b, v)
                if lo < hi:
                    dead[lo - a:hi - a] = False
            assert float(g[a:b][dead].abs().max()) == 0.0 if bool(dead.any()) else True
    assert plan["gradient_megabytes_per_step"] >= 4 * live / 1e6 and plan["ranks"] == 8
    # one per stop of the backward sweep + the 28 loss sums, + one where a stop's live words are split by a dead gap larger than
    # train.SPAN_GAP_MAX (round 6, ADVICE r5: 31 MB of grad-less zeros inside the first piece are no longer sent along to save a collective)
    assert plan["collectives_per_step"] <= 5, plan["collectives_per_step"]
    assert plan["pieces"][0]["sent_when_backward_reaches"] == "encoder_done" and plan["pieces"][-1]["sent_when_backward_reaches"].startswith("end")
    assert plan["total_ring_ms_one_link"] > plan["total_ring_ms_seven_links"] > 0
    live_params = sum(p.numel() for p in m.parameters() if p.grad is not None)
    assert live >= live_params                                    # buckets merge small gaps; nothing live is left out
