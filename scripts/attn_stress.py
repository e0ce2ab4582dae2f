"""Race hunt for the three hand-scheduled bridge-attention streams (forward, dQ, dK/dV).

The streams synchronise by counted waits and one workgroup barrier per few sub-tiles of an LDS ring; a missing wait or a slot reused a
barrier too early would show up only under some timing.  Their results involve no atomics, so any two runs on the same operands must agree
BIT FOR BIT: this script repeats forward + backward `--iters` times per shape, on fresh random operands every `--reseed` iterations, while a
side stream keeps the chip busy with unrelated traffic of changing size (to move the relative timing of workgroups), and compares every
result with the first one of its operand set; every operand set is also checked against an fp64 statement of the attention on a sample of
queries.  Exit code 1 on any mismatch.

    python scripts/attn_stress.py [--iters 400] [--reseed 50]
"""
import argparse
import ctypes as C
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from transception_amd._lib import lib, TC_BF16, TC_F16   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=400)
ap.add_argument("--reseed", type=int, default=50)
args = ap.parse_args()
L = lib()
dev = torch.device("cuda:0")
d = 64
SHAPES = [   # (B, queries per scale, keys, storage type): the bench shape, config 5 (384^2 fp16), config 4 (512^2), ragged tails, a short key set
    (16, (3136, 1568, 980, 392), 784, torch.bfloat16),
    (8, (9216, 4608, 2880, 1152), 2304, torch.float16),
    (8, (16384, 8192, 5120, 2048), 4096, torch.bfloat16),
    (3, (1000, 333, 65, 31), 200, torch.bfloat16),
    (2, (777, 100), 97, torch.float16),
]
side = torch.cuda.Stream()
noise_a = torch.randn(64 << 20, device=dev)
noise_b = torch.empty_like(noise_a)
bad = 0
for B, nq, Nk, dtype in SHAPES:
    dt = TC_BF16 if dtype == torch.bfloat16 else TC_F16
    rows = B * sum(nq)
    nqc = (C.c_int * len(nq))(*nq)
    scale = 0.125
    gen = torch.Generator(device=dev).manual_seed(1234 + rows)
    o = torch.empty(rows, d, device=dev, dtype=dtype)
    dq = torch.empty_like(o)
    dkv = torch.empty(B * Nk, 2 * d, device=dev, dtype=dtype)
    lse, delta = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    dkv32 = torch.empty(8 * B * Nk * 128, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ref = None
    mism = 0
    for it in range(args.iters):
        if it % args.reseed == 0:
            qf = torch.randn(rows, d, device=dev, generator=gen)
            q = (qf * (scale * 1.4426950408889634)).to(dtype)           # the model hands Q over as q * scale * log2(e), rounded once
            kv = torch.randn(B * Nk, 2 * d, device=dev, generator=gen).to(dtype)
            do = torch.randn(rows, d, device=dev, generator=gen).to(dtype)
            k, v = kv[:, :d], kv[:, d:]
            ref = None
        # unrelated traffic of a size that changes every iteration, on a second stream
        n = (1 + (it * 7919) % 61) << 20
        with torch.cuda.stream(side):
            noise_b[:n].copy_(noise_a[:n])
        o.fill_(float("nan")); dq.fill_(float("nan")); dkv.fill_(float("nan"))
        L.tc_attn_fwd_seg(q.data_ptr(), d, k.data_ptr(), 2 * d, v.data_ptr(), 2 * d, Nk * 2 * d, o.data_ptr(), d, lse.data_ptr(), B, len(nq), nqc, Nk,
                          scale, 1, dt, st)
        L.tc_attn_bwd_seg(q.data_ptr(), d, k.data_ptr(), 2 * d, v.data_ptr(), 2 * d, Nk * 2 * d, o.data_ptr(), d, do.data_ptr(), d, lse.data_ptr(),
                                delta.data_ptr(), dkv32.data_ptr(), dq.data_ptr(), d, dkv.data_ptr(), 2 * d, dkv[:, d:].data_ptr(), 2 * d, Nk * 2 * d, B,
                                len(nq), nqc, Nk, scale, 1, dt, st)
        got = (o.clone(), lse.clone(), dq.clone(), dkv.clone())
        if ref is None:
            ref = got
            # fp64 statement on a sample of queries of every scale
            off = 0
            for si, n_s in enumerate(nq):
                for b in (0, B - 1):
                    r0 = off + b * n_s
                    sel = torch.arange(r0, r0 + n_s, max(1, n_s // 37), device=dev)
                    s = (q[sel].double() @ k[b * Nk:(b + 1) * Nk].double().T) * 0.6931471805599453     # Q carries scale * log2(e)
                    p = torch.softmax(s, dim=1)
                    want = p @ v[b * Nk:(b + 1) * Nk].double()
                    err = (o[sel].double() - want).abs().max().item()
                    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
                    if not err < tol:
                        print(f"  FORWARD vs fp64: shape {B, nq, Nk} scale {si} image {b}: {err:.3e}")
                        bad += 1
                off += B * n_s
            if not all(torch.isfinite(t.float()).all() for t in got):
                print(f"  NON-FINITE result: shape {B, nq, Nk}")
                bad += 1
        else:
            for name, a, r in zip(("O", "lse", "dQ", "dK|dV"), got, ref):
                if not torch.equal(a, r):
                    ndiff = int((a != r).sum())
                    print(f"  MISMATCH {name}: shape {B, nq, Nk} {dtype} iteration {it}: {ndiff} elements differ, max {(a.float() - r.float()).abs().max().item():.3e}")
                    mism += 1
    torch.cuda.synchronize()
    bad += mism
    print(f"B={B} nq={nq} Nk={Nk} {str(dtype)[6:]}: {args.iters} runs, {args.iters // args.reseed} operand sets, run-to-run mismatches {mism}")
print("RACE HUNT", "FAILED" if bad else "clean")
sys.exit(1 if bad else 0)
