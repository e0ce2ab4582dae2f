import ctypes as C, sys, glob, os, torch
dev = torch.device("cuda:0")
B, nq, Nk, d = 16, [3136, 1568, 980, 392], 784, 64
rows = B * sum(nq)
torch.manual_seed(0)
q = torch.randn(rows, d, device=dev).bfloat16(); kv = torch.randn(B * Nk, 2 * d, device=dev).bfloat16()
lse = torch.empty(rows, device=dev)
nqc = (C.c_int * 4)(*nq); st = torch.cuda.current_stream().cuda_stream
k, v = kv[:, :d], kv[:, d:]
ref = None
here = os.path.dirname(os.path.abspath(__file__))
for so in sorted(glob.glob(here + "/libatt_v*.so"), key=lambda p: int(p.split("_v")[1][:-3])):
    L = C.CDLL(so)
    f = L.tc_attn_fwd_seg
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
    o = torch.zeros_like(q)
    def fwd(): return f(q.data_ptr(), d, k.data_ptr(), 2 * d, v.data_ptr(), 2 * d, Nk * 2 * d, o.data_ptr(), d, lse.data_ptr(), B, 4, nqc, Nk, 0.125, 1, st)
    rc = fwd(); torch.cuda.synchronize()
    for _ in range(5): fwd()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): fwd()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30
    of = o.float()
    if ref is None: ref = of
    err = (of - ref).abs().max().item()
    print(f"{os.path.basename(so):16s} rc={rc} {us:7.1f} us  {4.0*rows*Nk*d/us/1e6:7.1f} TF/s  max|o-o_v0|={err:.2e}")
