import os, sys, ctypes as C, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from transception_amd._lib import TC_BF16, TC_F16, lib
from transception_amd.engine import ATTN_DKV_SPLITS
from transception_amd.seeded_init import seeded_tensor
DEV = "cuda:0"
L = lib()
d, scale, l2e = 64, 0.125, 1.4426950408889634
dtype = torch.float16 if len(sys.argv) < 2 else getattr(torch, sys.argv[1])
dt = TC_BF16 if dtype == torch.bfloat16 else TC_F16
st = torch.cuda.current_stream().cuda_stream
T = lambda tag, shape: torch.from_numpy(seeded_tensor("ops/" + tag, shape, 1.0))
for ci, (B, nq, Nk) in enumerate([(1, [64], 64), (1, [33], 65), (2, [100, 37], 80), (2, [1, 32, 31], 95)]):
    rows = B * sum(nq)
    q = (T(f"dv.q{ci}", (rows, d)).to(DEV) * (scale * l2e)).to(dtype)
    kv = T(f"dv.kv{ci}", (B * Nk, 2 * d)).to(DEV).to(dtype)
    k, v = kv[:, :d], kv[:, d:]
    do = (T(f"dv.g{ci}", (rows, d)).to(DEV) * float(os.environ.get("DO_SCALE", "1"))).to(dtype)
    nqc = (C.c_int * 4)(*(list(nq) + [0] * (4 - len(nq))))
    o = torch.empty((rows, d), device=DEV, dtype=dtype); lse = torch.empty((rows,), device=DEV)
    L.tc_attn_fwd_seg(q.data_ptr(), d, k.data_ptr(), 2 * d, v.data_ptr(), 2 * d, Nk * 2 * d, o.data_ptr(), d, lse.data_ptr(), B, len(nq), nqc, Nk, scale, 1, dt, st)
    outs = {}
    for impl in ("1", "0", "1b", "0b"):
        os.environ["TC_ATTN_DKV_ASM"] = impl[0]
        dq = torch.zeros((rows, d), device=DEV).to(dtype); dkv = torch.zeros((B * Nk, 2 * d), device=DEV).to(dtype)
        delta = torch.empty((rows,), device=DEV); dkv32 = torch.zeros((ATTN_DKV_SPLITS * B * Nk * 128,), device=DEV)
        L.tc_attn_bwd_seg(q.data_ptr(), d, k.data_ptr(), 2 * d, v.data_ptr(), 2 * d, Nk * 2 * d, o.data_ptr(), d, do.data_ptr(), d, lse.data_ptr(), delta.data_ptr(),
                          dkv32.data_ptr(), dq.data_ptr(), d, dkv.data_ptr(), 2 * d, dkv.data_ptr() + 2 * d, 2 * d, Nk * 2 * d, B, len(nq), nqc, Nk, scale, 1, dt, st)
        torch.cuda.synchronize()
        outs[impl] = (dkv.float().cpu(), dkv32[:4 * B * Nk * 128].view(4, B, Nk, 128).cpu().clone())
    a, b_ = outs["1"][0], outs["0"][0]
    diff = (a - b_).abs()
    idx = diff.nonzero()
    print((B, nq, Nk), "ndiff", len(idx), "max", diff.max().item())
    p1, p0 = outs["1"][1], outs["0"][1]
    pd = (p1 - p0).abs()
    pi = pd.nonzero()
    print("  partial diffs", len(pi), "max", pd.max().item(), "first", pi[:6].tolist())
    print("  asm vs asm", (outs["1"][1] - outs["1b"][1]).abs().max().item(), " hip vs hip", (outs["0"][1] - outs["0b"][1]).abs().max().item())
