"""scripts/exp/gemm_glds.hip (LDS-DMA K loop) against the library's tc_gemm on forward-Linear shapes of the step: y = x W^T, bf16."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from transception_amd._lib import lib, TcGemm, TC_BF16
dev = torch.device("cuda:0")
P = C.CDLL(os.path.join(ROOT, "scripts", "exp", "libgemm_glds.so"))
P.gemm_glds.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
L = lib()

def time_us(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)

shapes = [(12544, 512, 128), (12544, 128, 512), (9408, 512, 128), (9408, 128, 512), (9408, 384, 128), (9408, 128, 128), (50176, 64, 64), (50176, 256, 64), (50176, 64, 256),
          (97216, 64, 64), (37632, 192, 64), (2352, 1280, 320), (2352, 320, 1280), (3136, 1280, 320), (784, 2048, 512), (784, 512, 2048), (50176, 1024, 64)]
print(f"{'M x N x K':>22s}  {'tc_gemm':>8s} {'glds d1':>8s} {'glds d2':>8s} {'glds d3':>8s}   floor(5TB/s)  max|diff|")
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    y0 = torch.empty(M, N, device=dev, dtype=torch.bfloat16); y1 = torch.empty_like(y0)
    g = TcGemm()
    g.A, g.B, g.C = x.data_ptr(), w.data_ptr(), y0.data_ptr()
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc = M, N, K, K, K, N
    g.transA, g.transB, g.nb1, g.nb2, g.splitk, g.alpha, g.dtype = 0, 1, 1, 1, 1, 1.0, TC_BF16
    def ref(): L.tc_gemm(C.byref(g), torch.cuda.current_stream().cuda_stream)
    ref(); torch.cuda.synchronize()
    res = []
    for d in (1, 2, 3):
        def run(d=d): P.gemm_glds(x.data_ptr(), K, w.data_ptr(), K, y1.data_ptr(), N, M, N, K, d, torch.cuda.current_stream().cuda_stream)
        y1.zero_(); run(); torch.cuda.synchronize()
        diff = float((y1.float() - y0.float()).abs().max())
        res.append((time_us(run), diff))
    t0 = time_us(ref)
    fl = (M * K + N * K + M * N) * 2 / 5e6
    print(f"{M:>8d}x{N:>5d}x{K:>5d}  {t0:8.1f} {res[0][0]:8.1f} {res[1][0]:8.1f} {res[2][0]:8.1f}   {fl:8.1f}     {max(r[1] for r in res):.3g}")
