"""Depthwise-conv microbenchmark (bf16): fwd / bwd-input / bwd-weight at the model's shapes, with the HBM-floor time."""
import sys, torch
sys.path.insert(0, "/root/repo")
from transception_amd._lib import lib, TC_BF16
L = lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
def t(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
wsp = torch.zeros(16384 + 1024 * 16384, dtype=torch.uint8, device=dev)
def run(B, H, W, C, k, groups=1):
    n = groups * B * H * W
    x = torch.randn(n, C, device=dev).bfloat16(); y = torch.empty_like(x); dy = torch.randn(n, C, device=dev).bfloat16(); dx = torch.empty_like(x)
    ws = C * k * k + C
    par = torch.randn(groups * ws, device=dev).bfloat16(); gpar = torch.zeros(groups * ws, device=dev)
    w, b = par.data_ptr(), par.data_ptr() + 2 * C * k * k
    f = t(lambda: L.tc_dwconv_fwd(x.data_ptr(), C, w, b, y.data_ptr(), C, B, H, W, C, k, 1, 1, groups, ws, TC_BF16, st))
    d = t(lambda: L.tc_dwconv_bwd_input(dy.data_ptr(), C, w, dx.data_ptr(), C, B, H, W, C, k, 1, 1, 0, groups, ws, TC_BF16, st))
    g = t(lambda: L.tc_dwconv_bwd_weight(dy.data_ptr(), C, x.data_ptr(), C, gpar.data_ptr(), gpar.data_ptr() + 4 * C * k * k, B, H, W, C, k, 1, groups, ws, wsp.data_ptr(), wsp.numel(), TC_BF16, st))
    floor = 2 * n * C * 2 / 8e6
    print(f"B={B} {H}x{W} C={C:4d} k={k} g={groups}: fwd {f:6.1f}  dx {d:6.1f}  dw {g:6.1f} us   (hbm floor {floor:5.1f} us)")
run(16, 56, 56, 256, 3); run(16, 28, 28, 512, 3); run(16, 14, 14, 1280, 3); run(16, 7, 7, 2048, 3)
run(16, 56, 56, 64, 3, 3); run(16, 56, 56, 64, 5, 3); run(16, 56, 56, 64, 7, 3)      # illustrative MHCA shapes
run(16, 56, 56, 16, 3, 3); run(16, 56, 56, 24, 5, 3); run(16, 56, 56, 24, 7, 3)
run(16, 28, 28, 128, 3, 3); run(16, 28, 28, 48, 7, 3); run(16, 14, 14, 320, 3, 3); run(16, 14, 14, 120, 7, 3)
