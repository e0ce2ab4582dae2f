#!/usr/bin/env python
"""bench.py -- images/sec, forward + backward (+ loss + SGD step), TransCeption 224x224, B=16 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype f32|bf16] [--batch 16] [--size 224] [--no-cpu]

N>1: launched by the driver as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...
bench.py --gpus N ...`; started WITHOUT a rendezvous environment, `--gpus N` re-executes itself through that very launcher, so a
multi-GPU figure can never be a silent one-GPU run (the world size is asserted and reported as n_gpus / rccl_ranks).

One "step" = one pass of the hot path over one batch of synthetic Synapse-shaped input resident in HBM: MSTransception forward,
0.4*CE+0.6*Dice loss, backward, (gradient all-reduce over RCCL when N>1), fused SGD update.  Weak scaling: 16 images per GPU.
`value` = images of the K timed steps / wall time between two barrier+synchronize brackets (max over ranks); the median of the
per-step HIP-event times is reported beside it (config.median_ms_per_step).  Rank 0 prints ONE JSON line carrying
  roofline      the bridge SR-attention forward kernel (the MFMA-bound kernel BASELINE.json's north_star names): HIP events around
                30 back-to-back launches inside a replayed hipGraph (the kernel's own duration; agrees with the rocprofv3 average);
                the per-launch event figure of instrumented eager steps (it includes the event-pair floor) is
                roofline_in_step_events;
  roofline_gemm the time-dominant GEMM family (pair / merged / single launches): algorithmic bytes AND flops per launch;
  roofline_step_dominant  the kernel with the largest share of the step (gemm_pair_kernel), same algorithmic work per launch over the
                duration rocprofv3 reports for it inside the replayed step (profiles/r4_step_timeline.json);
  step          the whole step: algorithmic TFLOP/s and fraction of the MFMA peak, memory-side GB/s from the committed PMC passes of
                one replayed step and traffic_over_algorithmic;
  roofline_hbm  the memory-bound family with the largest share of the step, timed the same way, with its PMC traffic;
  cpu_baseline  the CPU oracle (a port of the reference arithmetic; the reference's Python cannot travel) timed on this box's
                host cores on a bounded sample of the same workload (rank 0, N=1 only): median of 3 steps + the config-1 figure.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f16": 2500.0}        # MI355X dense MFMA peaks (MI355X_MICROARCH.md)
TORCH_DTYPE = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
LOSS_SCALE = {"f32": 1.0, "bf16": 1.0, "f16": 4096.0}     # float16 activations need a loss scale (gradients of order 1e-6)
PEAK_HBM_GBS = 8000.0


def _profile(*names):
    """The newest committed profile summary among `names` (profiles/<name>) -> (doc, file name, fresh).  fresh: its provenance stamp
    (scripts/provenance.py: sha256 of the kernel sources + host package it was collected from) equals the sources this run executes;
    a stale or unstamped profile still names its file, but bench reports its traffic as null with traffic_stale = true."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("provenance", os.path.join(ROOT, "scripts", "provenance.py"))
    prov = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(prov)
    for n in names:
        f = os.path.join(ROOT, "profiles", n)
        if os.path.exists(f):
            try:
                doc = json.load(open(f))
            except ValueError:
                continue
            return doc, n, prov.matches(doc, ROOT)
    return {}, None, False


def synthetic_batch(B: int, size: int, device, seed: int):
    """Synapse-shaped input: one-channel slice normalised to [-1, 1] (trainer.py:89-92) + integer labels in 0..8."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = (torch.rand(B, 1, size, size, generator=g) - 0.5) / 0.5
    y = torch.randint(0, 9, (B, size, size), generator=g)
    return x.to(device), y.to(device)


def _loader_feed(args, dev, rank: int, world: int, n_needed: int, out=None, raw=False):
    """Generator of (x, y) batches from the device input pipeline over a synthetic Synapse-format training set written to local
    disk (512x512 npz slices, dataset_synapse.py:103-107).  raw: (loader, generator of raw batch slots) instead -- the consumer
    captures the preprocessing launches at the head of its step graph (_loader_fed_step)."""
    import tempfile
    from transception_amd import data as D
    tmp = tempfile.mkdtemp(prefix=f"synapse_r{rank}_")
    D.write_synthetic_synapse(tmp + "/train_npz", tmp + "/lists", n_cases=4, slices_per_case=32, size=512, seed=1234)
    ds = D.SynapseSlices(tmp + "/train_npz", tmp + "/lists")
    per_epoch = (len(ds) // (args.batch * world)) * args.batch * world
    loader = D.DeviceLoader(ds, args.batch, img_size=args.size, device=dev, seed=1234, rank=rank, world=world, augment=True,
                            epochs=n_needed // per_epoch + 2, readers=8, out=None if raw else out)
    return (loader, loader.iter_raw(int(os.environ.get("TC_BENCH_LOADER_SLOTS", "3")))) if raw else iter(loader)


def _loader_fed_step(args, dev, rank, world, n_needed, make_step):
    """A step() fed by the device input pipeline with NOTHING eager between two step graphs: the raw batch (fp32 slices, uint8 labels,
    augmentation records) lands in one of three static device slots under the previous step, and the augmentation / spline / zoom
    launches are captured at the head of the step graph of that slot (one captured step per slot; make_step(pre) builds it)."""
    loader, it = _loader_feed(args, dev, rank, world, n_needed, raw=True)
    steps = {}

    def step():
        slot = next(it)
        st = steps.get(slot["index"])
        if st is None:
            st = steps[slot["index"]] = make_step(loader.slot_preprocess(slot))
        return st()
    return step, steps, it


# ----------------------------------------------------------------------------------------------------------------- CPU baseline
def _cpu_baseline_worker(batch: int, size: int):
    """Oracle on the host cores: config 1 (B=2 train-mode forward) and the benchmarked workload (fwd+bwd+SGD at the
    benchmarked batch: config 2 = B=16), each the median of 3 repetitions after a warm-up.  Bounded: about 4 steps of the full batch
    (~7 s each on a 2-socket host); only when a probe says they would take longer than ~90 s is the batch cut to 4 images (and the line
    says so).  Thread count: the faster of 32 / 64 threads on the warm-up probe (more threads than that only add synchronisation cost
    to this model's many small ops: 256 threads ran several times slower); physical / logical core counts of the host are stated."""
    from oracle.transception_oracle import TransCeptionOracle, ce_dice_loss, load_params
    from transception_amd.seeded_init import seeded_state_dict
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    P = load_params(seeded_state_dict(), requires_grad=True)
    leaves = list({id(v): v for v in P.values() if v.requires_grad}.values())
    opt = torch.optim.SGD(leaves, lr=0.05, momentum=0.9, weight_decay=1e-4)
    orc = TransCeptionOracle(P, 9, training=True)

    def fwd(x):
        t0 = time.perf_counter()
        with torch.no_grad():
            orc(x)
        return time.perf_counter() - t0

    def step(x, y):
        t0 = time.perf_counter()
        loss, _, _ = ce_dice_loss(orc(x), y, 9)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return time.perf_counter() - t0

    xp, yp = synthetic_batch(2, size, "cpu", 1)
    cands = sorted({c for c in (32, 64) if 1 <= c <= avail} or {avail})
    best, cores = None, cands[0]
    for c in cands:                                      # doubles as the warm-up
        torch.set_num_threads(c)
        fwd(xp)
        t = fwd(xp)
        if best is None or t < best:
            best, cores = t, c
    torch.set_num_threads(cores)
    f2 = statistics.median(fwd(xp) for _ in range(3))
    probe = step(xp, yp)
    bs = batch if probe * batch / 2 * 4 < 90.0 else min(batch, 4)
    x, y = synthetic_batch(bs, size, "cpu", 1)
    times = [step(x, y) for _ in range(3)]
    t = statistics.median(times)
    phys = _physical_cores()
    ratio = _reference_cpu_ratio()
    # SURVEY.md 8(d) words the baseline as "n = all physical host cores": that figure beside the fastest thread count (one step after one
    # warm-up step: the oversubscribed pool is several times slower, and the whole baseline has to stay inside its time budget)
    allp = None
    if phys and phys != cores and phys <= avail:
        torch.set_num_threads(phys)
        step(xp, yp)
        ta = step(x, y)
        allp = {"cores": phys, "value": bs / ta, "unit": "images/sec", "sample": f"one fwd+bwd+SGD step of B={bs} after a B=2 warm-up step"}
        torch.set_num_threads(cores)
    out = {"value": bs / t, "unit": "images/sec", "cores": cores, "cores_available": avail, "physical_cores": phys, "kind": "port",
           "all_physical_cores": allp,
           "batch": bs, "config1_b2_fwd_images_per_sec": 2 / f2,
           "sample": f"median of 3 fwd+bwd+SGD steps of B={bs} {size}x{size} after warm-up (and median of 3 train-mode forwards of B=2 = "
                     f"BASELINE config 1), fp32 PyTorch-CPU oracle (port of the reference arithmetic), {cores} threads "
                     f"(fastest of {cands} on the probe) of {avail} logical / {phys} physical cores"}
    if ratio is not None:
        out["reference_cpu_ratio"] = ratio
    return out


def _physical_cores():
    """Physical cores of the host: distinct (physical id, core id) pairs of /proc/cpuinfo (None when it cannot be read)."""
    try:
        seen, pid = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                seen.add((pid, line.split(":")[1].strip()))
        return len(seen) or None
    except OSError:
        return None


def _reference_cpu_ratio():
    """Oracle-vs-reference CPU timing measured in the build container (the reference's Python cannot travel to the GPU box) by
    scripts/time_reference_cpu.py and committed as profiles/reference_cpu_ratio.json: the port is FASTER than the reference, so the
    cpu_baseline above is a conservative (high) stand-in for the reference's own CPU rate."""
    f = os.path.join(ROOT, "profiles", "reference_cpu_ratio.json")
    try:
        return json.load(open(f))
    except (OSError, ValueError):
        return None


def cpu_baseline(batch: int, size: int, timeout: float = 300.0):
    """Runs the worker in a child process so a slow host can never stall the GPU measurement."""
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--batch", str(batch), "--size", str(size)],
                           capture_output=True, text=True, timeout=timeout)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "images/sec", "cores": 0, "kind": "port", "sample": "worker failed: " + r.stderr[-200:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "images/sec", "cores": 0, "kind": "port", "sample": f"worker exceeded {timeout:.0f} s"}


# ------------------------------------------------------------------------------------------------------------------ launcher
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _self_spawn(n: int) -> int:
    """`python bench.py --gpus N` without a rendezvous environment: re-run this command line under torch.distributed.run, one rank
    per GPU (the launch line the driver uses)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# --------------------------------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default="bf16", choices=["f32", "bf16", "f16"])
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-side", action="store_true", help="skip the side figures (forward-only, fp32, loader-fed, roofline passes)")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of replaying the captured hipGraph")
    ap.add_argument("--force-split", action="store_true", help="use the 3-graph multi-GPU step structure even on one GPU")
    ap.add_argument("--resident", action="store_true", help="time the step on ONE HBM-resident noise batch (the headline of rounds 1-4) instead of the "
                    "loader-fed step; the other figure is reported beside the headline either way")
    ap.add_argument("--loader", action="store_true",
                    help="feed the timed steps from transception_amd.data.DeviceLoader (synthetic Synapse npz files on local disk -> "
                         "HBM -> device augmentation/resize) instead of one HBM-resident batch; the default keeps BASELINE's definition")
    ap.add_argument("--comm-plan", type=int, default=8, metavar="N",
                    help="attach config.comm_plan: what a data-parallel step of N GPUs would send (pieces, buckets, expected xGMI ring time) -- "
                         "computed on this rank, nothing is sent")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        print(json.dumps(_cpu_baseline_worker(args.batch, args.size)))
        return

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(_self_spawn(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or drop the rendezvous environment and "
                         f"let `bench.py --gpus N` spawn its own ranks)")
    one_gpu_drill = os.environ.get("TC_TEST_ONE_GPU") == "1"    # drill of the multi-rank step on a 1-GPU box: every rank on cuda:0,
    if one_gpu_drill:                                           # collectives through TC_DIST_BACKEND=gloo (not a measurement)
        local = 0
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    group, backend = None, None
    rccl1 = world == 1 and args.force_split and not args.eager     # one GPU: the split step with its collectives on a 1-rank RCCL group
    if world > 1 or rccl1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rccl1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            import transception_amd.train as _tr
            _tr.COMM_AT_WORLD_1 = True
        backend = os.environ.get("TC_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        group = dist.group.WORLD
        assert dist.get_world_size() == args.gpus
    comm = world > 1 or rccl1

    import transception_amd.engine as engine
    from transception_amd import MSTransception
    from transception_amd.seeded_init import seeded_state_dict
    from transception_amd.train import FusedSGD, GraphedStep, SegLoss, cosine_lr, train_step

    def build_model(dtype: str):
        m = MSTransception(num_classes=9)
        m.load_state_dict(seeded_state_dict(), strict=True)      # random-init weights of the architecture (name-seeded)
        m.to(dev).train()
        m.set_compute_dtype(TORCH_DTYPE[dtype])
        m._ensure_flat(dev)
        return m

    model = build_model(args.dtype)
    if comm:
        dist.broadcast(model.flat_parameters(), src=0)           # C3: identical replicas
    loss_fn = SegLoss(9, group=group, loss_scale=LOSS_SCALE[args.dtype])
    opt = FusedSGD(model, lr=0.05, momentum=0.9, weight_decay=1e-4)
    x, y = synthetic_batch(args.batch, args.size, dev, 1234 + rank)
    t_max = max(args.steps + args.warmup, 1)

    def sync():
        torch.cuda.synchronize(dev)
        if comm:
            dist.barrier()
            torch.cuda.synchronize(dev)

    if args.eager:
        step = lambda: train_step(model, loss_fn, opt, x, y, group)
    else:
        step = GraphedStep(model, loss_fn, opt, x, y, group, warmup=2, force_split=args.force_split)   # capture, then replay
    feed = None
    # BASELINE config 2 words the workload as "synthetic Synapse npz": the timed region is the LOADER-FED step by default (npz slices on local
    # disk -> pinned staging -> HBM -> device augmentation / spline zoom captured at the head of the step graph); --resident times the step
    # on one HBM-resident batch instead (rounds 1-4's headline), and the other figure is reported beside the headline either way.
    if not args.resident and not args.eager:
        args.loader = True
    loader_note = None
    if args.loader and not args.eager:
        step_resident = step
        try:
            step, fed_steps, feed = _loader_fed_step(args, dev, rank, world, (args.steps + args.warmup + 4) * args.batch * world,
                                                     lambda pre: GraphedStep(model, loss_fn, opt, x, y, group, warmup=1, force_split=args.force_split, pre=pre))
            for _ in range(3):                                   # every slot's step graph is captured before the warm-up / timed steps
                step()
        except Exception as e:                                   # (a box without a writable temp dir, ...): say so and time the resident batch
            if world > 1:
                raise
            loader_note = f"loader-fed step unavailable ({type(e).__name__}: {e}); resident batch timed instead"
            step, feed, args.loader = step_resident, None, False
    for i in range(args.warmup):
        step()
        opt.set_lr(cosine_lr(0.05, i + 1, t_max))
    sync()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss, ce, dice = step()
        opt.set_lr(cosine_lr(0.05, args.warmup + i + 1, t_max))
        marks[i + 1].record()
    sync()
    elapsed = time.perf_counter() - t0
    per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    fed_launches = None
    if feed is not None:
        feed.close()
        fed_launches = next(iter(fed_steps.values())).kernel_nodes()
        step = step_resident
    exposed = None
    if comm:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if not args.eager:
            # the same captured step without its collectives: the difference is the all-reduce time the step does not hide
            k = min(args.steps, 20)
            sync()
            t1 = time.perf_counter()
            for _ in range(k):
                step(comm=False)
            sync()
            t = torch.tensor([(time.perf_counter() - t1) / k], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            exposed = 1e3 * (elapsed / args.steps - float(t.item()))
            dist.broadcast(model.flat_parameters(), src=0)       # the ranks' un-reduced updates diverged: not used afterwards
            model.invalidate_working_copy()

    launches = None if args.eager else (fed_launches or (step.kernel_nodes() if hasattr(step, "kernel_nodes") else None))
    extra, roofs = {}, {}
    if rank == 0 and world == 1 and not args.no_side:
        extra, roofs = side_figures(args, dev, model, loss_fn, opt, step, x, y, build_model, engine, train_step, GraphedStep, FusedSGD, SegLoss,
                                     extra_step_seconds=(elapsed / args.steps,))

    if rank == 0:
        out = {
            "metric": "images/sec fwd+bwd at 224x224 B=16 per GPU", "value": world * args.batch * args.steps / elapsed,
            "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype,
            "data": ("synthetic (one HBM-resident batch of uniform-noise slices in [-1, 1] with random labels; config.loader_fed_images_per_sec "
                     "is the same step fed from synthetic Synapse npz files)" if not args.loader else
                     "synthetic Synapse npz files (512x512 smooth-blob slices) through DeviceLoader: read, H2D, augment, resize inside the timed region; "
                     "config.resident_batch_images_per_sec is the same step on one HBM-resident batch") + (f" [{loader_note}]" if loader_note else ""),
            "config": {"workload": f"TransCeption (MSTransception) {args.size}x{args.size} B={args.batch}/GPU fwd+bwd+SGD, "
                                   + ("random-noise batch resident in HBM" if not args.loader else "synthetic Synapse npz slices via the device loader")
                                   + ", name-seeded random-init weights",
                       "global_batch": world * args.batch, "image_size": args.size, "parallelism": f"dp{world}",
                       #  which workload `value` timed, machine-readable: "loader" (the default since round 5), "resident"
                       # (--resident / --eager; rounds 1-4's headline), and degraded = the loader-fed step was asked for and could not run
                       "headline_mode": "loader" if args.loader else "resident", "headline_degraded": bool(loader_note),
                       "launch_mode": "eager" if args.eager else "hipGraph replay", "final_loss": float(loss.item()),
                       "launches_per_step": launches,
                       "median_ms_per_step": statistics.median(per_step), "min_ms_per_step": min(per_step), **extra},
        }
        if comm:
            out["rccl_ranks"] = dist.get_world_size()
            out["config"]["collective_backend"] = backend + (" (all ranks on one GPU: drill, not a measurement)" if one_gpu_drill else "") + \
                (" (one rank: the split step's collectives run on a 1-rank communicator -- the N=1 floor of allreduce_exposed_ms)" if rccl1 else "")
            out["config"]["allreduce_exposed_ms"] = exposed
        if args.comm_plan > 1:
            from transception_amd.train import comm_plan
            out["config"]["comm_plan"] = comm_plan(model, max(world, args.comm_plan) if world == 1 else world)
        out.update(roofs)
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.batch, args.size)
        print(json.dumps(out))
    if comm:
        # Every rank is done (barrier + device sync), then the process leaves WITHOUT the C++ / C-stdio teardown: RCCL writes a version
        # banner through C stdio that would be flushed to stdout after the JSON line, and tearing the communicator down while its
        # watchdog thread is alive has aborted the process once (SIGABRT after a complete, correct run).
        sync()
        sys.stdout.flush()
        sys.stderr.flush()
        if os.environ.get("TC_BENCH_SOFT_EXIT", "0") != "1":         # (a tracer needs the ordinary teardown to write its files)
            os._exit(0)


ATTN_KERNEL = ("attn_fwd_asm_kernel (bridge SR-attention forward, QK^T + softmax + PV fused, all 4 scales x B images in one launch; the "
               "hand-scheduled gfx950 stream of csrc/gen_attn_asm.py -- TC_ATTN_FWD_ASM=0 selects the compiler-scheduled attn_fwd_seg_kernel)")


def _graph_replay_us(fn, n: int, dev, settle_s: float = 0.3, timed_s: float = 0.1, burst: dict | None = None) -> float:
    """Average duration of fn's launches when n of them run back-to-back inside one replayed hipGraph (HIP events on the stream
    the graph is launched on), in the STEADY state: the graph is replayed for `settle_s` seconds first and then timed over
    enough replays to fill `timed_s`.  Round 6 measured why (scripts/exp/attn_power.py, profiles/r6_attn_fwd_sustained_power.txt):
    the MFMA streams run against the package power limit (~1300 W, sclk ~2.1 GHz on random operands), and a single 30-launch
    replay after an idle gap (what rounds 2-5 timed) is over before the clock has settled -- it read 25.6 us for a kernel that
    sustains 22.6-24.2 us.  The single-replay figure is still taken and returned through `burst` for continuity."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record()
    torch.cuda.synchronize(dev)
    one = e0.elapsed_time(e1) * 1e-3                      # seconds per replay, cold clocks
    if burst is not None:
        burst["single_replay_us"] = one * 1e6 / n
    if settle_s <= 0:
        return one * 1e6 / n
    chunk = max(1, int(0.02 / max(one, 1e-6)))            # ~20 ms of replays between host synchronisations
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < settle_s:
        for _ in range(chunk):
            g.replay()
        torch.cuda.synchronize(dev)
    reps = max(1, int(timed_s / max(one, 1e-6)))
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) * 1e3 / (n * reps)


def side_figures(args, dev, model, loss_fn, opt, step, x, y, build_model, engine, train_step, GraphedStep, FusedSGD, SegLoss, extra_step_seconds=()):
    """Rank 0, one GPU, after the timed region: figures SURVEY.md 8(d) asks for beside the headline."""
    extra, roofs = {}, {}
    from transception_amd._lib import _Lib
    # (1) instrumented steps of the same workload: HIP events around the launches of the named kernels (engine.PROFILE)
    engine.PROFILE = {}
    for _ in range(3):
        train_step(model, loss_fn, opt, x, y, None)
    torch.cuda.synchronize(dev)
    prof, engine.PROFILE = engine.PROFILE, None
    # what an event pair around a launch reports for a kernel that does nothing, on an idle queue (the instrumented steps are eager and
    # host-bound, so every timed launch starts on an idle GPU): dispatch + event overhead included in every in-step figure below
    from transception_amd._lib import lib as _libf
    _L, _one = _libf(), torch.zeros(64, device=dev)
    _fl = []
    for _ in range(24):
        torch.cuda.synchronize(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _L.tc_fill_f32(_one.data_ptr(), 64, 0.0, torch.cuda.current_stream().cuda_stream)
        b.record()
        torch.cuda.synchronize(dev)
        _fl.append(a.elapsed_time(b) * 1e3)
    event_floor_us = statistics.median(_fl[4:])
    c0 = _Lib.calls
    train_step(model, loss_fn, opt, x, y, None)
    extra["c_abi_calls_per_step"] = _Lib.calls - c0
    torch.cuda.synchronize(dev)
    peak = PEAK_TFLOPS[args.dtype]

    def mfma_block(ev, kernel):
        ms = sum(a.elapsed_time(b) for a, b, _ in ev)
        fl = sum(f for _, _, f in ev)
        ach = fl / (ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": kernel, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                "launches": len(ev), "avg_launch_us": 1e3 * ms / len(ev), "algorithmic_flops_per_launch": fl / len(ev),
                "event_pair_floor_us": event_floor_us,
                "how": "HIP events around each launch inside 3 instrumented (eager) training steps of the benchmarked workload; an event "
                       "pair around an empty kernel on the idle queue reads event_pair_floor_us, which every in-step figure includes -- the "
                       "kernel's own duration is roofline_graph_replay / the rocprofv3 average in profiles/"}
    # committed profile summaries of this workload; a summary counts only when its provenance stamp matches the sources this run executes
    pmc_all, prof_file, pmc_fresh = _profile("r6_hbm_by_kernel.json", "r5_hbm_by_kernel.json", "r4_hbm_by_kernel.json", "r3_hbm_by_kernel.json", "r2_hbm_by_kernel.json")
    pmc_doc = pmc_all.get("kernels", {})
    tl_all, tl_file, tl_fresh = _profile("r6_step_timeline.json", "r5_step_timeline.json", "r4_step_timeline.json", "r3_step_timeline.json")
    tl_doc = tl_all.get("kernels", {})
    same_workload = args.dtype == "bf16" and args.batch == 16 and args.size == 224
    STALE = "stale: the kernel sources changed since this profile was collected (scripts/provenance.py); re-run scripts/profile_round5.sh"

    def pmc_traffic(prefixes):
        """Memory-side bytes per launch of the kernels whose names start with one of `prefixes`, from the committed rocprofv3 --pmc
        passes of this workload (2 x FETCH_SIZE + WRITE_SIZE, scripts/pmc_step.sh); None for another workload."""
        if not same_workload:
            return None, None
        if not pmc_fresh:
            return None, f"profiles/{prof_file}: {STALE}" if prof_file else None
        hit = [v for k, v in pmc_doc.items() if any(k.startswith(q) for q in prefixes)]
        n = sum(v["launches_per_step"] for v in hit)
        if not n:
            return None, None
        return sum(v["bytes_per_step"] for v in hit) / n, (f"profiles/{prof_file} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE passes of this workload, "
                                                            "scripts/pmc_step.sh; per launch; collected from the sources this run executes)")
    if prof.get("attn_fwd"):
        r = mfma_block(prof["attn_fwd"], ATTN_KERNEL)
        r["traffic"], r["traffic_source"] = None, None
        adoc, aname, afresh = _profile("r6_attn_pmc.json", "r5_attn_pmc.json", "r4_attn_pmc.json", "r2_attn_pmc.json", "r1_attn_pmc.json")   # HBM bytes per launch: a separate rocprofv3 --pmc pass, committed
        if same_workload and aname:
            ent = adoc.get("attn_fwd_asm_kernel") or adoc.get("attn_fwd_seg_kernel") or {}
            if afresh:
                r["traffic"] = ent.get("hbm_bytes_corrected")
                r["traffic_source"] = f"profiles/{aname} (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE pass of this workload; collected from the sources this run executes)"
            else:
                r["traffic_stale"], r["traffic_source"] = True, f"profiles/{aname}: {STALE}"
        roofs["roofline_in_step_events"] = r
    if prof.get("attn_bwd"):
        roofs["roofline_attn_bwd"] = mfma_block(prof["attn_bwd"], "attn_bwd (dQ stream incl. the row deltas + dK/dV stream + partial fold of the bridge SR-attention)")
    # the GEMM family (the time-dominant one): algorithmic flops AND bytes per launch kind
    PMC_GEMM = {"single": ("gemm_bf16_kernel", "gemm_kernel"), "pair": ("gemm_pair_kernel",), "multi": ("gemm_multi_kernel",)}
    gem = []
    for kind in ("pair", "multi", "single"):
        ev = prof.get("gemm:" + kind)
        if not ev:
            continue
        ms = sum(a.elapsed_time(b) for a, b, _ in ev)
        fl = sum(w[0] for _, _, w in ev)
        by = sum(w[1] for _, _, w in ev)
        tr, src = pmc_traffic(PMC_GEMM[kind])
        gem.append({"bound": "hbm", "kernel": {"pair": "gemm_pair_kernel (dX + dW of a Linear in one grid)", "multi": "gemm_multi_kernel (<= 12 independent "
                    "Linear-type problems in one grid)", "single": "gemm_bf16_kernel / gemm_kernel (one problem per launch)"}[kind],
                    "achieved": by / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": by / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                    "achieved_tflops": fl / (ms * 1e-3) / 1e12, "frac_of_mfma_peak": fl / (ms * 1e-3) / 1e12 / peak,
                    "launches": len(ev), "launches_per_step": len(ev) / 3.0, "avg_launch_us": 1e3 * ms / len(ev), "ms_per_step": ms / 3.0,
                    "algorithmic_bytes_per_launch": by / len(ev), "algorithmic_flops_per_launch": fl / len(ev),
                    "traffic": tr, "traffic_source": src, "traffic_stale": bool(same_workload and prof_file and not pmc_fresh), "event_pair_floor_us": event_floor_us})
    if gem:
        gem.sort(key=lambda r: -r["ms_per_step"])
        roofs["roofline_gemm"] = dict(gem[0], how="HIP events around each GEMM-family launch inside 3 instrumented (eager) steps; the products are K = 64..2048 "
                                      "deep, i.e. HBM-bound (bound = hbm: algorithmic operand + result bytes / time), the MFMA figure is given beside it; every "
                                      "in-step figure includes event_pair_floor_us per launch")
        roofs["roofline_gemm_others"] = gem[1:]
        # the step-dominant kernel beside the named (attention) one: the same algorithmic work per launch over the launch duration
        # rocprofv3 reports for it inside the replayed step (no event-pair floor), from the committed timeline of this workload
        dom = dict(gem[0])
        names = PMC_GEMM[{"gemm_pair_kernel": "pair", "gemm_multi_kernel": "multi"}.get(dom["kernel"].split(" ")[0], "single")]
        hit = [v for k, v in tl_doc.items() if any(k.startswith(q) for q in names)] if (same_workload and tl_fresh) else []
        if same_workload and tl_file and not tl_fresh:
            dom["duration_stale"], dom["duration_source"] = True, f"profiles/{tl_file}: {STALE} (the in-step event figure above stands in)"
        if hit:
            n_l, ms_l = sum(v["launches"] for v in hit), sum(v["ms"] for v in hit)
            us = 1e3 * ms_l / n_l
            dom.update(avg_launch_us=us, launches_per_step=n_l, ms_per_step=ms_l, share_of_step=ms_l / max(1e-9, sum(v["ms"] for v in tl_doc.values())),
                       achieved=dom["algorithmic_bytes_per_launch"] / (us * 1e-6) / 1e9, duration_source=f"profiles/{tl_file} (rocprofv3 --kernel-trace of one replayed step)")
            dom["frac"] = dom["achieved"] / PEAK_HBM_GBS
            dom["achieved_tflops"] = dom["algorithmic_flops_per_launch"] / (us * 1e-6) / 1e12
            dom["frac_of_mfma_peak"] = dom["achieved_tflops"] / peak
            if dom.get("traffic"):
                dom["traffic_over_algorithmic"] = dom["traffic"] / dom["algorithmic_bytes_per_launch"]
        dom.pop("event_pair_floor_us", None)
        roofs["roofline_step_dominant"] = dom
    PMC_HBM = {"ffn_mid_bwd": ("ffn_mid_bwd_kernel",), "ffn_fused_fwd": ("ffn_fused_fwd_kernel",), "ffn_fused_bwd": ("ffn_bwd_",), "ffn_dw_fwd": ("dw_tile_kernel", "dw_multi_kernel"),
               "layernorm_fwd": ("ln_fwd_kernel",), "layernorm_bwd": ("ln_bwd_kernel",), "dwconv_fwd": ("dw_tile_kernel", "dw_multi_kernel<bf16, 0>", "dw_kernel"),
               "dwconv_bwd_input": ("dw_tile_kernel<bf16, 3, 8, 1>", "dw_multi_kernel<bf16, 1>"), "dwconv_bwd_weight": ("dw_tile_wgrad_kernel", "dw_multi_wgrad_kernel", "dw_wgrad_kernel"),
               "batchnorm_fwd": ("bn_partial_kernel<bf16, 0>", "bn_apply_kernel"), "batchnorm_bwd": ("bn_partial_kernel<bf16, 1>", "bn_bwd_apply_kernel"),
               "effatt_fwd": ("effatt_kv_kernel", "effatt_ctx_kernel", "effatt_out_kernel"), "effatt_bwd": ("effatt_bq_kernel", "effatt_dctx_kernel", "effatt_bkv_kernel", "effatt_fold_kernel"),
               "coord_pool_fwd": ("coord_pool_fwd_kernel",), "coord_gate_fwd": ("coord_gate_fwd_kernel",), "pixel_shuffle": ("pixel_shuffle",)}
    hbm = []
    for name, ev in prof.items():
        if name.startswith("hbm:") and ev:
            ms = sum(a.elapsed_time(b) for a, b, _ in ev)
            by = sum(f for _, _, f in ev)
            gbs = by / (ms * 1e-3) / 1e9
            tr, src = pmc_traffic(PMC_HBM.get(name[4:].split(" ")[0], ("\0",)))
            hbm.append({"bound": "hbm", "kernel": name[4:], "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                        "launches": len(ev), "avg_launch_us": 1e3 * ms / len(ev), "algorithmic_bytes_per_launch": by / len(ev),
                        "traffic": tr, "traffic_source": src, "traffic_stale": bool(same_workload and prof_file and not pmc_fresh)})
    if hbm:
        hbm.sort(key=lambda r: -r["avg_launch_us"] * r["launches"])
        roofs["roofline_hbm"] = dict(hbm[0], how="HIP events around each launch inside 3 instrumented steps; algorithmic bytes = every operand read once + "
                                     "every result written once (the memory-bound family with the largest share of the step); traffic = memory-side bytes per "
                                     "launch of the family's kernels from the committed PMC passes")
        roofs["roofline_hbm_others"] = hbm[1:6]
    # whole step: algorithmic work (16.88 GFLOP forward per 224^2 image, backward = 2x; fused-unit bytes 19.6 MB of bf16
    # activations per image and 92.7 MB of weights per step, x3 for forward + backward) over the measured step, and the memory-side
    # traffic of one replayed step from the committed PMC passes
    if args.size == 224 and args.dtype in ("bf16", "f16"):
        step_s = extra_step_seconds[0] if extra_step_seconds else None
        if step_s:
            fl = 3 * 16.88e9 * args.batch
            alg_b = 3 * (19.6e6 * args.batch + 92.7e6)
            st = {"algorithmic_flops_per_step": fl, "achieved_tflops": fl / step_s / 1e12, "frac_of_mfma_peak": fl / step_s / 1e12 / peak,
                  "algorithmic_bytes_per_step": alg_b}
            if same_workload and prof_file and not pmc_fresh:
                st.update(traffic_bytes_per_step=None, traffic_stale=True, traffic_source=f"profiles/{prof_file}: {STALE}")
            if same_workload and pmc_fresh and pmc_all.get("step_hbm_GB"):
                tb = 1e9 * pmc_all["step_hbm_GB"]
                st.update(traffic_bytes_per_step=tb, achieved_GBps=tb / step_s / 1e9, frac_of_hbm_peak=tb / step_s / 1e9 / PEAK_HBM_GBS,
                          traffic_over_algorithmic=tb / alg_b, traffic_source=f"profiles/{prof_file} (2 x FETCH_SIZE + WRITE_SIZE over one replayed step)")
            # the launch-bound share as a number: what a step of this many launches costs when every launch is an EMPTY kernel -- the
            # per-boundary cost measured here (200 empty one-wave kernels back to back in one replayed hipGraph) times the step's launches
            try:
                from transception_amd._lib import lib as _lf
                _Lm = _lf()
                bus = _graph_replay_us(lambda: _Lm.tc_seg_marker(0, torch.cuda.current_stream(dev).cuda_stream), 200, dev)
                nl = step.kernel_nodes() if hasattr(step, "kernel_nodes") else None
                if nl:
                    st.update(launch_boundary_us=bus, launches=nl, launch_floor_ms=bus * nl * 1e-3, launch_floor_share_of_step=bus * nl * 1e-6 / step_s)
            except Exception as e:                               # (never let a side figure take the headline down)
                st["launch_floor_error"] = repr(e)
            roofs["step"] = st
    # (2) forward-only rate (train-mode forward captured alone)
    with torch.no_grad():
        us = _graph_replay_us(lambda: model(x), 10, dev)
        extra["fwd_only_images_per_sec"] = args.batch / (us * 1e-6)
    # (3) the fp32 parity path on the same workload (the path whose results are pinned to the reference at 1e-4)
    if args.dtype != "f32" and not args.eager:
        m32 = build_model("f32")
        o32 = FusedSGD(m32, lr=0.05, momentum=0.9, weight_decay=1e-4)
        s32 = GraphedStep(m32, SegLoss(9), o32, x, y, None, warmup=2)
        for _ in range(3):
            s32()
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        for _ in range(10):
            s32()
        torch.cuda.synchronize(dev)
        extra["fp32_images_per_sec"] = args.batch * 10 / (time.perf_counter() - t)
        del s32, m32, o32
    # (4) the same graphed step fed by the device input pipeline (npz -> HBM -> augment/resize), SURVEY 8(f)-1
    if not args.eager and args.loader:
        for _ in range(3):
            step()
        torch.cuda.synchronize(dev)
        wins = []
        for _ in range(3):                               # the captured step on ONE HBM-resident batch (rounds 1-4's headline): three windows of 20 steps
            tl = time.perf_counter()
            for _ in range(20):
                step()
            torch.cuda.synchronize(dev)
            wins.append(args.batch * 20 / (time.perf_counter() - tl))
        extra["resident_batch_images_per_sec"] = statistics.median(wins)
        extra["resident_batch_ms_per_step"] = 1e3 * args.batch / statistics.median(wins)
        extra["resident_batch_windows_images_per_sec"] = wins
        extra["resident_batch_launches_per_step"] = step.kernel_nodes() if hasattr(step, "kernel_nodes") else None
    if not args.eager and not args.loader and args.size == 224:
        fstep, fsteps, f2 = _loader_fed_step(args, dev, 0, 1, 70 * args.batch,
                                              lambda pre: GraphedStep(model, loss_fn, opt, x, y, None, warmup=1, pre=pre))
        for _ in range(6):                               # the first three calls capture the three slot graphs
            fstep()
        torch.cuda.synchronize(dev)
        wins = []
        for _ in range(3):                               # three windows of 20 steps: the median is reported, all three are listed
            tl = time.perf_counter()
            for _ in range(20):
                fstep()
            torch.cuda.synchronize(dev)
            wins.append(args.batch * 20 / (time.perf_counter() - tl))
        extra["loader_fed_images_per_sec"] = statistics.median(wins)
        extra["loader_fed_windows_images_per_sec"] = wins
        extra["loader_fed_launches_per_step"] = next(iter(fsteps.values())).kernel_nodes()
        f2.close()
        del fsteps, fstep
    # (5) the roofline kernel back-to-back inside a replayed hipGraph on step-shaped operands (the micro-benchmark: warm caches and
    #     clocks; the rocprofv3 average of this command mixes it with the in-step launches)
    if args.dtype in ("bf16", "f16") and args.size % 32 == 0:
        import ctypes as C
        from transception_amd._lib import TC_BF16, TC_F16, lib
        tcd = TC_BF16 if args.dtype == "bf16" else TC_F16
        Bq, S = args.batch, args.size
        sides = [S // 4, S // 8, S // 16, S // 32]
        nq = [sides[i] * sides[i] * m_ for i, m_ in enumerate((1, 2, 5, 8))]
        Nk = (sides[3] * sides[3]) * (1 + 2 + 5 + 8)
        rows = Bq * sum(nq)
        q = torch.randn(rows, 64, device=dev).to(TORCH_DTYPE[args.dtype]); kv = torch.randn(Bq * Nk, 128, device=dev).to(TORCH_DTYPE[args.dtype])
        o = torch.empty_like(q); lse = torch.empty(rows, device=dev)
        nqc = (C.c_int * 4)(*nq)
        L = lib()

        def attn():
            L.tc_attn_fwd_seg(q.data_ptr(), 64, kv.data_ptr(), 128, kv[:, 64:].data_ptr(), 128, Nk * 128, o.data_ptr(), 64, lse.data_ptr(),
                              Bq, 4, nqc, Nk, 0.125, 1, tcd, torch.cuda.current_stream(dev).cuda_stream)   # qscaled = 1: as the model calls it
        burst_f = {}
        us = _graph_replay_us(attn, 30, dev, burst=burst_f)
        fl = 4.0 * rows * Nk * 64
        ins = roofs.get("roofline_in_step_events", {})
        roofs["roofline"] = {"bound": "mfma", "kernel": ATTN_KERNEL,
                             "achieved": fl / us / 1e6, "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                             "frac": fl / us / 1e6 / PEAK_TFLOPS[args.dtype], "avg_launch_us": us, "algorithmic_flops_per_launch": fl,
                             "single_cold_replay_us": burst_f.get("single_replay_us"),
                             "traffic": ins.get("traffic"), "traffic_source": ins.get("traffic_source"), "traffic_stale": bool(ins.get("traffic_stale")),
                             "how": "HIP events on the launching stream around back-to-back launches of the kernel (30 per replayed hipGraph), step-shaped "
                                    "RANDOM operands, steady state: 0.3 s of replays first, then 0.1 s timed (the stream runs against the package power "
                                    "limit, ~2.1 GHz at ~1300 W; single_cold_replay_us is the one replay after an idle gap that rounds 2-5 reported; on "
                                    "all-zero operands the same kernel holds 2.4 GHz at ~990 W and takes ~19.8 us: profiles/r6_attn_fwd_sustained_power.txt). "
                                    "The per-launch event figure of instrumented eager steps, which includes event_pair_floor_us, is roofline_in_step_events"}
        roofs["roofline_graph_replay"] = dict(roofs["roofline"], note="the same figure as `roofline` under a name that says how it is taken (the key `roofline` "
                                              "carried the in-step event figure until round 2; that one is roofline_in_step_events)")
        # the backward of the same call, the same way (dQ stream -- it also makes the row deltas -- dK/dV stream, partial fold)
        from transception_amd.engine import ATTN_DKV_SPLITS
        do = torch.randn(rows, 64, device=dev).to(TORCH_DTYPE[args.dtype])
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        delta = torch.empty(rows, device=dev); dkv32 = torch.empty(ATTN_DKV_SPLITS * Bq * Nk * 128, device=dev)

        def attn_bwd():
            L.tc_attn_bwd_seg(q.data_ptr(), 64, kv.data_ptr(), 128, kv[:, 64:].data_ptr(), 128, Nk * 128, o.data_ptr(), 64, do.data_ptr(), 64, lse.data_ptr(),
                              delta.data_ptr(), dkv32.data_ptr(), dq.data_ptr(), 64, dkv.data_ptr(), 128, dkv[:, 64:].data_ptr(), 128, Nk * 128, Bq, 4, nqc, Nk,
                              0.125, 1, tcd, torch.cuda.current_stream(dev).cuda_stream)
        burst_b = {}
        usb = _graph_replay_us(attn_bwd, 20, dev, burst=burst_b)
        flb = 10.0 * rows * Nk * 64
        roofs["roofline_attn_bwd_graph_replay"] = {
            "bound": "mfma", "kernel": "tc_attn_bwd_seg: attn_bwd_dq_asm_kernel (S, dP, dQ products + the row deltas) + attn_bwd_dkv_asm_kernel (S, dP, dV, dK "
                                       "products) + attn_dkv_store_kernel; hand-scheduled streams of csrc/gen_dq_asm.py / gen_dkv_asm.py",
            "achieved": flb / usb / 1e6, "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s", "frac": flb / usb / 1e6 / PEAK_TFLOPS[args.dtype],
            "avg_launch_us": usb, "algorithmic_flops_per_launch": flb, "single_cold_replay_us": burst_b.get("single_replay_us"),
            "frac_counting_4_products": 0.8 * flb / usb / 1e6 / PEAK_TFLOPS[args.dtype],
            "how": "as roofline: 20 back-to-back calls inside one replayed hipGraph; flops = the five products a backward without stored "
                   "probabilities needs (S, dP, dV, dQ, dK; frac_counting_4_products leaves S out); 7 products are executed (each stream "
                   "recomputes S and dP)"}
    elif "roofline_in_step_events" in roofs:
        roofs["roofline"] = roofs["roofline_in_step_events"]
    # (6) the memory-bound stages north_star names, alone: RIPM (Patch_Embed_stage) and IFF (CoordAtt), forward + backward of the three
    #     encoder stages of each in one replayed graph (scripts/bench_stage.py); memory-side bytes from the committed PMC passes of the same command
    if args.dtype == "bf16" and args.size == 224 and not args.eager:
        import importlib.util
        spec = importlib.util.spec_from_file_location("bench_stage", os.path.join(ROOT, "scripts", "bench_stage.py"))
        bs = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bs)
        pmc, pname, pfresh = _profile("r6_ripm_iff_hbm.json", "r5_ripm_iff_hbm.json", "r4_ripm_iff_hbm.json")
        if args.batch != 16:
            pmc, pname = {}, None
        for which, key in (("ripm", "roofline_ripm"), ("iff", "roofline_iff")):
            _, one_pass = bs.build(which, args.batch, dev)
            us = _graph_replay_us(one_pass, 4, dev)
            by = 3.0 * bs.ELEMS[which] * args.batch * 2
            tr = pmc.get(which, {}).get("traffic_bytes_per_pass") if pfresh else None
            roofs[key] = {"bound": "hbm", "kernel": {"ripm": "RIPM: Patch_Embed_stage x 3 encoder stages (dw3x3 + pw1x1 + BatchNorm + Hardswish, x 3 per stage), forward + backward",
                                                      "iff": "IFF: CoordAtt x 3 encoder stages (pool, conv1 + BatchNorm + act, conv_h / conv_w + sigmoid, gate, conv_in_out), forward + backward"}[which],
                          "achieved": by / us / 1e3, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": by / us / 1e3 / PEAK_HBM_GBS, "us_per_pass": us,
                          "algorithmic_bytes_per_pass": by, "traffic": tr, "traffic_GBps": tr / us / 1e3 if tr else None,
                          "launches_per_pass": pmc.get(which, {}).get("launches_per_pass") if pfresh else None,
                          "traffic_stale": bool(pname and not pfresh),
                          "traffic_source": (f"profiles/{pname} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE passes of scripts/bench_stage.py, scripts/pmc_stage.sh; collected "
                                             "from the sources this run executes)" if tr else (f"profiles/{pname}: {STALE}" if pname else None)),
                          "how": "the stage functions of the model on random maps of the step's shapes, captured once and replayed (HIP events on the launching "
                                 "stream); algorithmic bytes = SURVEY.md 8(d): every activation of the fused unit read / written once, backward = 2 x forward"}
    return extra, roofs


if __name__ == "__main__":
    main()
