"""Checks tc_attn_fwd_seg (hand-scheduled stream by default; TC_ATTN_FWD_ASM=0 selects the compiler-scheduled kernel) against an
fp64 torch statement of softmax(Q K^T scale) V on the same 16-bit inputs, over ragged / tail / re-reference cases, and times it."""
import ctypes as C, sys, os, torch
sys.path.insert(0, "/root/repo")
from transception_amd._lib import lib, TC_BF16, TC_F16
L = lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
d = 64

QSCALED = int(os.environ.get("QSCALED", "1"))   # 1: Q handed over as q * scale * log2(e), rounded once (what the model does)
LOG2E = 1.4426950408889634


def run(B, nq, Nk, dtype, scale=0.125, spike=False, seed=0, time_it=False):
    g = torch.Generator(device=dev).manual_seed(seed)
    rows = B * sum(nq)
    tdt = torch.bfloat16 if dtype == TC_BF16 else torch.float16
    qf = torch.randn(rows, d, device=dev, generator=g)
    q = (qf * (scale * LOG2E)).to(tdt) if QSCALED else qf.to(tdt)
    kv = torch.randn(B * Nk, 2 * d, device=dev, generator=g).to(tdt)
    if spike:                                     # late keys that outgrow the first sub-tile's reference exponent by far more than 2^30
        kvf = kv.float().view(B, Nk, 2 * d)
        kvf[:, Nk // 2 + 3, :d] *= 40.0
        kvf[:, Nk - 1, :d] *= 90.0
        kv = kvf.view(B * Nk, 2 * d).to(tdt)
    k, v = kv[:, :d], kv[:, d:]
    o = torch.full((rows, d), float("nan"), device=dev).to(tdt)
    lse = torch.full((rows,), float("nan"), device=dev)
    nqc = (C.c_int * 4)(*(list(nq) + [0] * (4 - len(nq))))
    def f(): return L.tc_attn_fwd_seg(q.data_ptr(), d, k.data_ptr(), 2 * d, v.data_ptr(), 2 * d, Nk * 2 * d, o.data_ptr(), d, lse.data_ptr(), B, len(nq), nqc, Nk, scale, QSCALED, dtype, st)
    rc = f(); torch.cuda.synchronize()
    # reference (stage-major query rows: segment s holds B images of nq[s] rows)
    ref_o = torch.empty(rows, d, dtype=torch.float64, device=dev); ref_l = torch.empty(rows, dtype=torch.float64, device=dev)
    off = 0
    kd, vd = k.double().view(B, Nk, d), v.double().view(B, Nk, d)
    for n in nq:
        qq = q[off: off + B * n].double().view(B, n, d)
        s = torch.einsum("bqd,bkd->bqk", qq, kd) * (1.0 / LOG2E if QSCALED else scale)   # the reference starts from the STORED operands
        ref_l[off: off + B * n] = torch.logsumexp(s, -1).reshape(-1)
        ref_o[off: off + B * n] = torch.einsum("bqk,bkd->bqd", torch.softmax(s, -1), vd).reshape(-1, d)
        off += B * n
    eo = (o.double() - ref_o).abs().max().item(); el = (lse.double() - ref_l).abs().max().item()
    rel = ((o.double() - ref_o).norm() / ref_o.norm()).item()
    msg = f"B={B} nq={nq} Nk={Nk} {'bf16' if dtype == TC_BF16 else 'f16'}{' spike' if spike else ''}: rc={rc} max|dO|={eo:.3e} relL2={rel:.3e} max|dlse|={el:.3e} nan={int(torch.isnan(o.float()).sum())}"
    if time_it:
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 30
        msg += f"  {us:.1f} us  {4.0 * rows * Nk * d / us / 1e6:.0f} TFLOP/s"
    print(msg, flush=True)
    return eo, el

print("TC_ATTN_FWD_ASM =", os.environ.get("TC_ATTN_FWD_ASM", "(default 1)"), " QSCALED =", QSCALED)
run(1, [64], 64, TC_BF16)
run(1, [33], 100, TC_BF16)
run(2, [100, 37], 64, TC_BF16)
run(2, [100, 37, 5], 96, TC_F16)
run(3, [392, 980], 784, TC_BF16, spike=True)
run(2, [392], 784, TC_F16, spike=True)
run(16, [3136, 1568, 980, 392], 784, TC_BF16, time_it=True)
run(16, [3136, 1568, 980, 392], 784, TC_F16, time_it=True)
run(8, [9216, 2304, 576, 144], 2304, TC_F16, time_it=True)
