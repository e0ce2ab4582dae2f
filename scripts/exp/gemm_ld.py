"""Does a power-of-two row pitch cost the bf16 GEMM HBM channels?  Same problem, operand rows padded by `pad` elements."""
import ctypes as C, sys, torch
sys.path.insert(0, "/root/repo")
from transception_amd._lib import lib, TcGemm, TC_BF16
L = lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
def run(M, N, K, pa=0, pc=0, iters=50, nb=1):
    a = torch.randn(nb * M, K + pa, device=dev).bfloat16()
    b = torch.randn(N, K, device=dev).bfloat16()
    c = torch.zeros(nb * M, N + pc, device=dev, dtype=torch.bfloat16)
    g = TcGemm(a.data_ptr(), b.data_ptr(), c.data_ptr(), None, None, M, N, K, K + pa, K, N + pc, 0, 0, 1, nb, 1,
               M * (K + pa), 0, 0, 0, M * (N + pc), 0, 0, 0, 1.0, 0, 0, 1, TC_BF16, 0, 0, None, 0, 0, 0, 0)
    for _ in range(5): L.tc_gemm(C.byref(g), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): L.tc_gemm(C.byref(g), st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    gb = (M * K + N * K + M * N) * 2.0 * nb / us / 1e3
    print(f"M={M:6d} N={N:4d} K={K:4d} nb={nb} padA={pa:3d} padC={pc:3d}: {us:7.1f} us {gb:7.0f} GB/s")
for nb in (1, 3):
    for pa in (0, 8, 32, 64):
        run(50176, 64, 256, pa=pa, nb=nb)
    for pc in (0, 8, 32, 64):
        run(50176, 256, 64, pc=pc, nb=nb)
    for pa, pc in ((0, 0), (8, 0), (0, 8), (8, 8)):
        run(12544, 128, 512, pa=pa, pc=pc, nb=nb)
