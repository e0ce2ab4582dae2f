"""The bridge MixFFN fc1 / fc2 launches (4 scales in one tc_gemm_multi) problem by problem and merged, in different orders.
python scripts/bench_multi.py"""
import ctypes as C, os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transception_amd._lib import lib, TcGemm, TC_BF16
L = lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
B = 16
scales = [(B * 3136, 64), (B * 784, 128), (B * 196, 320), (B * 49, 512)]
WS = 64 << 20
ws = torch.zeros(WS, dtype=torch.uint8, device=dev)

def problems(kind):
    out = []
    for i, (M, Cc) in enumerate(scales):
        K, N = (Cc, 4 * Cc) if kind == "fc1" else (4 * Cc, Cc)
        x = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
        y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        g = TcGemm(x.data_ptr(), w.data_ptr(), y.data_ptr(), b.data_ptr(), None, M, N, K, K, K, N, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1.0, 0, 0, 1, TC_BF16, 0, 0, None, 0, 0,
                   0, 0, ws.data_ptr() + i * (WS // 4), WS // 4)
        out.append((g, (x, w, b, y), 2.0 * M * N * K, 2.0 * (M * K + N * K + M * N)))
    return out

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

for kind in ("fc1", "fc2"):
    P = problems(kind)
    tot = 0
    for i, (g, keep, fl, by) in enumerate(P):
        us = timeit(lambda: L.tc_gemm(C.byref(g), st))
        tot += us
        print(f"{kind} scale {i}: M={g.M} N={g.N} K={g.K}: {us:6.1f} us  {fl/us/1e6:6.1f} TF/s  {by/us/1e6:5.2f} TB/s")
    print(f"{kind} sum of separate launches {tot:.1f} us")
    for order in ((0, 1, 2, 3), (3, 2, 1, 0), (2, 3, 1, 0), (1, 0, 3, 2)):
        arr = (TcGemm * 4)(*[P[i][0] for i in order])
        us = timeit(lambda: L.tc_gemm_multi(arr, 4, st))
        print(f"{kind} merged order {order}: {us:6.1f} us   {sum(p[2] for p in P)/us/1e6:6.1f} TF/s  {sum(p[3] for p in P)/us/1e6:5.2f} TB/s")
