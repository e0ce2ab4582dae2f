"""Two ranks on ONE MI355X (gloo collectives on device tensors): the captured multi-GPU step -- forward graph, loss-sum all-reduce,
backward cut at the encoder mark with the bridge/decoder buckets reduced asynchronously, encoder backward, SGD graph -- must follow
the plain eager step with one synchronous gradient all-reduce.  (RCCL itself needs two GPUs; the driver's scaling run covers it.)"""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from transception_amd import MSTransception
        from transception_amd.seeded_init import seeded_input, seeded_labels, seeded_state_dict
        from transception_amd.train import FusedSGD, GraphedStep, SegLoss, gradient_buckets, split_buckets, train_step
        dev = torch.device("cuda:0")
        x = torch.from_numpy(seeded_input(4))[2 * rank:2 * rank + 2].to(dev)
        y = torch.from_numpy(seeded_labels(4))[2 * rank:2 * rank + 2].to(dev)

        def fresh():
            m = MSTransception(num_classes=9)
            m.load_state_dict(seeded_state_dict(), strict=True)
            m.to(dev).train()
            m.set_compute_dtype(torch.float32)
            return m
        me, mg = fresh(), fresh()
        oe, og = FusedSGD(me, lr=0.05), FusedSGD(mg, lr=0.05)
        le, lg = SegLoss(9, group=dist.group.WORLD), SegLoss(9, group=dist.group.WORLD)
        eager = [float(train_step(me, le, oe, x, y, dist.group.WORLD)[0]) for _ in range(2)]      # = the capture's two warm-up steps
        step = GraphedStep(mg, lg, og, x, y, dist.group.WORLD, warmup=2)
        assert step.split and step.distributed
        late = mg.late_gradient_offset()
        early_b, late_b = split_buckets(gradient_buckets(mg), late)
        assert early_b and late_b and all(b <= late for _, b in early_b) and all(a >= late for a, _ in late_b)
        diffs = []
        for _ in range(3):
            a = float(train_step(me, le, oe, x, y, dist.group.WORLD)[0])
            b = float(step()[0])
            diffs.append(abs(a - b))
        pe, pg = me.flat_parameters(), mg.flat_parameters()
        ret[rank] = (max(diffs), float((pe - pg).abs().max()), eager[0])
    finally:
        dist.destroy_process_group()


def test_two_rank_captured_step_follows_eager():
    world = 2
    ret = mp.Manager().dict()
    port = 29600 + (os.getpid() % 300)
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == 2
    for r in range(world):
        dl, dp, first = ret[r]
        assert dl < 2e-5, dict(ret)                          # same losses step by step (fp32 compute)
        assert dp < 1e-4, dict(ret)                          # and the same parameters after 5 steps
    assert abs(ret[0][2] - ret[1][2]) < 1e-7                 # both ranks form the loss of the global batch


def test_bench_self_spawns_two_ranks_through_the_launcher():
    """`python bench.py --gpus 2` with no rendezvous environment must start two ranks itself (torch.distributed.run) and say so:
    n_gpus = rccl_ranks = 2, global batch doubled.  Here both ranks share the one GPU and the collectives go through gloo
    (TC_TEST_ONE_GPU=1: a drill of the launcher and of the split-graph step, not a measurement)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TC_TEST_ONE_GPU="1", TC_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "2",
                        "--no-cpu"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["config"]["global_batch"] == 4
    assert out["config"]["allreduce_exposed_ms"] is not None and out["value"] > 0


def test_bench_refuses_a_world_size_mismatch():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
