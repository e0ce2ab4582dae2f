"""Micro-benchmark of the bridge SR-attention launches at the bench shape (B=16, 224^2): HIP-event timing in the steady state (round 6:
the streams run against the package power limit, a 30-launch burst after an idle gap is over before the clock settles -- TC_BENCH_ITERS launches
back to back, default 2000 forward / 600 backward after a quarter of that as warm-up; TC_BENCH_ITERS=30 restores the burst of rounds 2-5)."""
import ctypes as C, sys, torch
sys.path.insert(0, "/root/repo")
from transception_amd._lib import lib, TC_BF16, TC_F32
L = lib()
dev = torch.device("cuda:0")
import os
_sc = float(os.environ.get("TC_BENCH_NQ_SCALE", "1"))          # < 1: fewer queries per image (separates the per-sub-tile cost of the streams from their fixed part)
B, nq, Nk, d = 16, [int(n * _sc) for n in (3136, 1568, 980, 392)], 784, 64
rows = B * sum(nq)
dtype = torch.bfloat16 if "--f32" not in sys.argv else torch.float32
dt = TC_BF16 if dtype == torch.bfloat16 else TC_F32
q = torch.randn(rows, d, device=dev).to(dtype); kv = torch.randn(B * Nk, 2 * d, device=dev).to(dtype)
o = torch.empty_like(q); do = torch.randn(rows, d, device=dev).to(dtype)
dq = torch.empty_like(q); dkv = torch.empty_like(kv)
lse = torch.empty(rows, device=dev); delta = torch.empty(rows, device=dev); dkv32 = torch.empty(8 * B * Nk * 128, device=dev)   # TC_ATTN_DKV_SPLITS partial buffers
nqc = (C.c_int * 4)(*nq)
QS = 0 if ("--unscaled" in sys.argv or dtype == torch.float32) else 1   # 1: Q handed over as q * scale * log2(e) (what the model does: the hand-scheduled forward)
st = torch.cuda.current_stream().cuda_stream
k, v = kv[:, :d], kv[:, d:]
def fwd(): L.tc_attn_fwd_seg(q.data_ptr(), d, k.data_ptr(), 2 * d, v.data_ptr(), 2 * d, Nk * 2 * d, o.data_ptr(), d, lse.data_ptr(), B, 4, nqc, Nk, 0.125, QS, dt, st)
def bwd(): L.tc_attn_bwd_seg(q.data_ptr(), d, k.data_ptr(), 2 * d, v.data_ptr(), 2 * d, Nk * 2 * d, o.data_ptr(), d, do.data_ptr(), d, lse.data_ptr(), delta.data_ptr(), dkv32.data_ptr(),
                             dq.data_ptr(), d, dkv.data_ptr(), 2 * d, dkv[:, d:].data_ptr(), 2 * d, Nk * 2 * d, B, 4, nqc, Nk, 0.125, QS, dt, st)
for name, fn, fl in (("fwd", fwd, 4.0 * rows * Nk * d), ("bwd", bwd, 10.0 * rows * Nk * d)):
    it = int(os.environ.get("TC_BENCH_ITERS", "2000" if name == "fwd" else "600"))
    for _ in range(max(5, it // 4)): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / it
    print(f"attn {name} {dtype}: {us:8.1f} us/launch  {fl / us / 1e6:8.1f} TFLOP/s")
