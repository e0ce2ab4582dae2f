"""As gemm_glds_bench.py, but every launch of a replayed graph reads operands no launch before it touched within ~600 MB (a pool of buffers larger
than L2 + the Infinity Cache): the K loop against HBM latency, as most GEMMs of a training step meet it."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from transception_amd._lib import lib, TcGemm, TC_BF16
dev = torch.device("cuda:0")
P = C.CDLL(os.path.join(ROOT, "scripts", "exp", "libgemm_glds.so"))
P.gemm_glds.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
L = lib()
shapes = [(12544, 128, 512), (9408, 128, 512), (9408, 512, 128), (2352, 320, 1280), (784, 512, 2048), (3136, 1280, 320), (50176, 64, 256), (9408, 128, 128)]
print(f"{'M x N x K':>22s}  {'tc_gemm':>8s} {'glds d1':>8s} {'glds d2':>8s} {'glds d3':>8s}   (cold operands)")
for M, N, K in shapes:
    per = (M * K + N * K + M * N) * 2
    npool = max(8, int(700e6 // per))
    xs = [torch.randn(M, K, device=dev).bfloat16() for _ in range(npool)]
    ws = [(torch.randn(N, K, device=dev) * K ** -0.5).bfloat16() for _ in range(npool)]
    ys = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(npool)]
    gs = []
    for i in range(npool):
        g = TcGemm()
        g.A, g.B, g.C = xs[i].data_ptr(), ws[i].data_ptr(), ys[i].data_ptr()
        g.M, g.N, g.K, g.lda, g.ldb, g.ldc = M, N, K, K, K, N
        g.transA, g.transB, g.nb1, g.nb2, g.splitk, g.alpha, g.dtype = 0, 1, 1, 1, 1, 1.0, TC_BF16
        gs.append(g)
    def timeit(launch):
        st = lambda: torch.cuda.current_stream().cuda_stream
        for i in range(npool): launch(i, st())
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for i in range(npool): launch(i, st())
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): gr.replay()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (3 * npool)
    t0 = timeit(lambda i, s: L.tc_gemm(C.byref(gs[i]), s))
    td = [timeit(lambda i, s, d=d: P.gemm_glds(xs[i].data_ptr(), K, ws[i].data_ptr(), K, ys[i].data_ptr(), N, M, N, K, d, s)) for d in (1, 2, 3)]
    print(f"{M:>8d}x{N:>5d}x{K:>5d}  {t0:8.1f} {td[0]:8.1f} {td[1]:8.1f} {td[2]:8.1f}   pool {npool}")
    del xs, ws, ys
