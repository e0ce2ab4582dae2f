import ctypes as C, sys, torch
sys.path.insert(0, "/root/repo")
from transception_amd._lib import lib, TcGemm, TC_BF16
L = lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
def run(M, N, K, tA=0, tB=1, c_f32=0, splitk=1, nb=1, iters=50):
    a = torch.randn((K, M) if tA else (M, K), device=dev).bfloat16().repeat(nb, 1)
    b = torch.randn((N, K) if tB else (K, N), device=dev).bfloat16().repeat(nb, 1)
    c = torch.zeros(nb * M, N, device=dev, dtype=torch.float32 if c_f32 else torch.bfloat16)
    g = TcGemm(a.data_ptr(), b.data_ptr(), c.data_ptr(), None, None, M, N, K, a.stride(0), b.stride(0), N, 0, tA, tB, nb, 1,
               a.shape[0] // nb * a.stride(0), 0, b.shape[0] // nb * b.stride(0), 0, M * N, 0, 0, 0, 1.0, int(splitk > 1), 0, splitk, TC_BF16, c_f32, 0, None, 0, 0, 0, 0)
    for _ in range(5): L.tc_gemm(C.byref(g), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): L.tc_gemm(C.byref(g), st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"M={M:6d} N={N:5d} K={K:6d} tA={tA} tB={tB} nb={nb} splitk={splitk:3d}: {us:7.1f} us  {2.0*M*N*K*nb/us/1e6:8.1f} TFLOP/s")
run(12544, 256, 64); run(12544, 64, 256); run(3136, 512, 128); run(3136, 128, 512); run(784, 1280, 320); run(784, 320, 1280)
run(9408, 512, 128, nb=3); run(50176, 256, 64); run(50176, 64, 256); run(802816, 1024 // 16, 64)
run(12544, 64, 256, tB=0)                                  # dX
run(256, 64, 12544, tA=1, tB=0, c_f32=1, splitk=8)         # dW
run(512, 128, 3136, tA=1, tB=0, c_f32=1, splitk=12)
run(1280, 320, 784, tA=1, tB=0, c_f32=1, splitk=3)
