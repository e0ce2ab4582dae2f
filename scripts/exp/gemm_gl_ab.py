"""tc_gemm on the step's backward shapes (dX = dY W: TA=0 TB=0; dW = dY^T X: TA=1 TB=0, fp32 atomics, split K) with cold operands;
run once with TC_GEMM_GLDS=1 and once with 0 to A/B the LDS-DMA K loop inside the library."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from transception_amd._lib import lib, TcGemm, TC_BF16
dev = torch.device("cuda:0"); L = lib()
def splitk(m_out, n_out, k_red):
    tiles = ((m_out + 63) // 64) * ((n_out + 63) // 64)
    return max(1, min(max(512, 128) // max(tiles, 1), k_red // 384, 128))
cases = [("dX", 9408, 128, 512), ("dX", 9408, 512, 128), ("dX", 12544, 128, 512), ("dX", 9408, 128, 384), ("dX", 2352, 320, 1280), ("dX", 50176, 64, 256),
         ("dW", 9408, 128, 512), ("dW", 9408, 512, 128), ("dW", 9408, 128, 128), ("dW", 12544, 128, 512), ("dW", 37632, 64, 64), ("dW", 50176, 64, 256), ("dW", 97216, 64, 64)]
print("TC_GEMM_GLDS =", os.environ.get("TC_GEMM_GLDS", "1"))
for kind, rows, cin, cout in cases:
    per = rows * (cin + cout) * 2
    npool = max(6, int(600e6 // per))
    dys = [torch.randn(rows, cout, device=dev).bfloat16() for _ in range(npool)]
    xs = [torch.randn(rows, cin, device=dev).bfloat16() for _ in range(npool)]
    w = (torch.randn(cout, cin, device=dev) * cout ** -0.5).bfloat16()
    dxs = [torch.empty(rows, cin, device=dev, dtype=torch.bfloat16) for _ in range(npool)]
    dw = torch.zeros(cout, cin, device=dev)
    gs = []
    for i in range(npool):
        g = TcGemm()
        if kind == "dX":
            g.A, g.B, g.C = dys[i].data_ptr(), w.data_ptr(), dxs[i].data_ptr()
            g.M, g.N, g.K, g.lda, g.ldb, g.ldc = rows, cin, cout, cout, cin, cin
            g.transA, g.transB, g.splitk = 0, 0, 1
        else:
            g.A, g.B, g.C = dys[i].data_ptr(), xs[i].data_ptr(), dw.data_ptr()
            g.M, g.N, g.K, g.lda, g.ldb, g.ldc = cout, cin, rows, cout, cin, cin
            g.transA, g.transB, g.c_f32, g.accumulate, g.splitk = 1, 0, 1, 1, splitk(cout, cin, rows)
        g.nb1, g.nb2, g.alpha, g.dtype = 1, 1, 1.0, TC_BF16
        gs.append(g)
    st = lambda: torch.cuda.current_stream().cuda_stream
    for i in range(npool): L.tc_gemm(C.byref(gs[i]), st())
    torch.cuda.synchronize()
    if kind == "dW":
        ref = sum(dys[i].float().t() @ xs[i].float() for i in range(npool))
        err = float((dw - ref).abs().max() / ref.abs().max())
    else:
        err = float((dxs[0].float() - dys[0].float() @ w.float()).abs().max())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(npool): L.tc_gemm(C.byref(gs[i]), st())
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): gr.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{kind} rows {rows:6d} Cin {cin:4d} Cout {cout:4d} splitk {gs[0].splitk:3d}: {e0.elapsed_time(e1) * 1e3 / (3 * npool):7.1f} us   err {err:.2e}")
    del dys, xs, dxs
