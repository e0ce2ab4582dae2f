"""The library's bf16 GEMM against torch.matmul (hipBLASLt / rocBLAS) on the small-M, deep products of the bridge / stage-3-4 MixFFNs:
is there headroom in the kernel itself at these shapes?  (graph-replayed back-to-back launches: no host gaps in either figure)"""
import ctypes as C, sys, torch
sys.path.insert(0, "/root/repo")
from transception_amd._lib import lib, TcGemm, TC_BF16
L = lib(); dev = torch.device("cuda:0")

def timed(fn, iters=30):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn(s.cuda_stream)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters): fn(s.cuda_stream)
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s); g.replay(); e1.record(s); s.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

def run(M, N, K, tA=0, tB=1, c_f32=0, splitk=1):
    a = torch.randn((K, M) if tA else (M, K), device=dev).bfloat16()
    b = torch.randn((N, K) if tB else (K, N), device=dev).bfloat16()
    c = torch.zeros(M, N, device=dev, dtype=torch.float32 if c_f32 else torch.bfloat16)
    g = TcGemm(a.data_ptr(), b.data_ptr(), c.data_ptr(), None, None, M, N, K, a.stride(0), b.stride(0), N, 0, tA, tB, 1, 1,
               0, 0, 0, 0, M * N, 0, 0, 0, 1.0, int(splitk > 1), 0, splitk, TC_BF16, c_f32, 0, None, 0, 0, 0, 0)
    mine = timed(lambda st: L.tc_gemm(C.byref(g), st))
    A = a.t() if tA else a
    Bm = b.t() if tB else b
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    blas = timed(lambda st: torch.matmul(A, Bm, out=o))
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:5d} K={K:6d} tA={tA} tB={tB}: library {mine:7.1f} us {fl / mine / 1e6:7.1f} TF   torch.matmul {blas:7.1f} us {fl / blas / 1e6:7.1f} TF")

for M, C_ in ((784, 512), (3136, 320), (12544, 128)):
    run(M, 4 * C_, C_)                 # fc1
    run(M, C_, 4 * C_)                 # fc2
    run(M, C_, 4 * C_, tB=0)           # dX of fc1:  dh [M, 4C] W1 [4C, C]
    run(M, 4 * C_, C_, tB=0)           # dX of fc2
    run(4 * C_, C_, M, tA=1, tB=0, c_f32=1, splitk=max(1, M // 1024))    # dW1 = dh^T x
    run(C_, 4 * C_, M, tA=1, tB=0, c_f32=1, splitk=max(1, M // 1024))    # dW2 = dy^T a
run(50176, 256, 64); run(50176, 64, 256)
