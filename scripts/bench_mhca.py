"""The fused attention half of an MHCABlock (tc_mhca_att_fwd) at the three encoder-stage shapes of the B=16 224^2 step: time per launch
(back-to-back launches between HIP events) against the three launches it replaces; with a -DTC_MHCA_TIMING library (TC_LIB_PATH) also
the per-phase cycles of a sample of workgroups.
    python scripts/bench_mhca.py [--timing]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import transception_amd.engine as E
from transception_amd.engine import Graph, P, Var
from transception_amd._lib import lib, TC_BF16

dev = torch.device("cuda:0")
timing = "--timing" in sys.argv
WINDOWS = [(3, 2), (5, 3), (7, 3)]
L = lib()
for C, side in ((64, 28), (128, 14), (320, 7)):
    B, Gn, Ch, N = 16, 3, C // 8, side * side
    rows = Gn * B * N
    torch.manual_seed(0)
    x = torch.randn(rows, C, device=dev).to(torch.bfloat16)
    shapes = [(3 * C, C), (3 * C,)] + [s for k, nh in WINDOWS for s in ((nh * Ch, k * k), (nh * Ch,))]
    sizes = [(int(np.prod(s)) + 7) // 8 * 8 for s in shapes]
    per = sum(sizes)
    flat = (0.1 * torch.randn(Gn * per, device=dev)).to(torch.bfloat16)
    gflat = torch.zeros(Gn * per, device=dev)
    Ps, off = [], 0
    for s, n in zip(shapes, sizes):
        ne = int(np.prod(s)); Ps.append(P(flat[off:off + ne].view(s), gflat[off:off + ne].view(s), per)); off += n

    def run(fused, iters=50):
        E._MHCA_ATT_FUSED = fused
        xv = Var(x)
        def once():
            G = Graph(torch.bfloat16, dev, training=True, record=False)       # (binds the current stream: inside a capture, the capture stream)
            with G.grouped(Gn, per):
                if fused:
                    return G.mhca_attention(xv, Ps[0], Ps[1], [Ps[2], Ps[4], Ps[6]], [Ps[3], Ps[5], Ps[7]], B, side, 8, Ch ** -0.5, WINDOWS)
                qkv = G.linear(xv, Ps[0], Ps[1])
                q, k, v = qkv.colslice(0, C), qkv.colslice(C, 2 * C), qkv.colslice(2 * C, 3 * C)
                convv = G.new(rows, C)
                c0, xs, outs = 0, [], []
                for ks, nh in WINDOWS:
                    xs.append(v.colslice(c0, c0 + nh * Ch)); outs.append(convv.colslice(c0, c0 + nh * Ch)); c0 += nh * Ch
                G.dwconv_multi(xs, [Ps[2], Ps[4], Ps[6]], [Ps[3], Ps[5], Ps[7]], (B, side, side), [3, 5, 7], outs)
                return G.factor_att_core(q, k, v, convv, Gn * B, N, 8, Ch ** -0.5)
        for _ in range(3):
            once()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters):
                once()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters
    tf, tu = run(True), run(False)
    print(f"C={C:3d} {side}x{side} B=16 x 3 paths: fused {tf:6.1f} us   op-by-op (3 launches) {tu:6.1f} us")
    gy = torch.randn(rows, C, device=dev).to(torch.bfloat16)

    def run_fb(fused_bwd, iters=30):
        E._MHCA_ATT_FUSED, E._MHCA_ATT_BWD_FUSED = True, fused_bwd
        xv = Var(x)
        def once():
            G = Graph(torch.bfloat16, dev, training=True, record=True)
            with G.grouped(Gn, per):
                o = G.mhca_attention(xv, Ps[0], Ps[1], [Ps[2], Ps[4], Ps[6]], [Ps[3], Ps[5], Ps[7]], B, side, 8, Ch ** -0.5, WINDOWS)
                o.root.grad_t = gy; o.root.whole_written = True
                G.backward()
            xv.grad_t = None; xv.whole_written = False; xv.written = []
        for _ in range(3):
            once()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters):
                once()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters
    if "--bwd" in sys.argv:
        a, b = run_fb(True), run_fb(False)
        print(f"       forward + backward (incl. the projection's gradient pair and the deferred folds): fused backward {a:6.1f} us   factor_att_bwd + dwconv_multi {b:6.1f} us")
    E._MHCA_ATT_FUSED = E._MHCA_ATT_BWD_FUSED = True
    if timing:
        Bt = Gn * B
        nst = Bt * 8 * 2 * Ch
        stats = torch.zeros(nst + Bt * 8 * 32, device=dev)
        qkv = torch.empty(rows, 3 * C, device=dev, dtype=torch.bfloat16); cv = torch.empty(rows, C, device=dev, dtype=torch.bfloat16); o = torch.empty_like(cv)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            L.tc_mhca_att_fwd(x.data_ptr(), C, Ps[0].data.data_ptr(), Ps[1].data.data_ptr(), Ps[2].data.data_ptr(), Ps[3].data.data_ptr(), Ps[4].data.data_ptr(),
                              Ps[5].data.data_ptr(), Ps[6].data.data_ptr(), Ps[7].data.data_ptr(), per, qkv.data_ptr(), 3 * C, cv.data_ptr(), C, o.data_ptr(), C,
                              stats.data_ptr(), Gn, B, side, side, C, Ch ** -0.5, TC_BF16, st)
        torch.cuda.synchronize()
        t = stats[nst:].cpu().numpy().view(np.int64).reshape(Bt * 8, 16)
        d = np.diff(t[:, :7], axis=1).astype(np.float64)
        names = ["qkv gemm", "qkv store", "conv", "softmax", "gram", "out"]
        print("   per-phase mean cycles by window:")
        for wname, hs in (("3x3", (0, 1)), ("5x5", (2, 3, 4)), ("7x7", (5, 6, 7))):
            m = np.isin(t[:, 7], hs)
            print(f"    {wname}  [projection phase: loads issued + taps parked {(t[m, 9] - t[m, 0]).mean():6.0f}  tiles {(t[m, 1] - t[m, 9]).mean():6.0f}]")
            print("    " + wname + "  " + "  ".join(f"{n} {d[m, i].mean():7.0f}" for i, n in enumerate(names)) + f"  total {(t[m, 6] - t[m, 0]).mean():7.0f}")

if "--timing-bwd" in sys.argv:
    for C, side in ((64, 28), (128, 14), (320, 7)):
        B, Gn, Ch, N = 16, 3, C // 8, side * side
        rows, Bt = Gn * B * N, Gn * B
        torch.manual_seed(0)
        shapes = [(3 * C, C), (3 * C,)] + [s for k, nh in WINDOWS for s in ((nh * Ch, k * k), (nh * Ch,))]
        sizes = [(int(np.prod(s)) + 7) // 8 * 8 for s in shapes]
        per = sum(sizes)
        flat = (0.1 * torch.randn(Gn * per, device=dev)).to(torch.bfloat16); gflat = torch.zeros(Gn * per, device=dev)
        offs = np.cumsum([0] + sizes)
        qkv = torch.randn(rows, 3 * C, device=dev).to(torch.bfloat16); cv = torch.randn(rows, C, device=dev).to(torch.bfloat16); go = torch.randn(rows, C, device=dev).to(torch.bfloat16)
        dqkv = torch.empty_like(qkv)
        nst = Bt * 8 * 2 * Ch
        stats = torch.ones(nst + Bt * 8 * 16, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        wp = lambda i: flat.data_ptr() + 2 * int(offs[i]); gp = lambda i: gflat.data_ptr() + 4 * int(offs[i])
        for _ in range(3):
            L.tc_mhca_att_bwd(qkv.data_ptr(), 3 * C, cv.data_ptr(), C, go.data_ptr(), C, stats.data_ptr(), dqkv.data_ptr(), 3 * C, 0, 0, 0, wp(2), wp(4), wp(6),
                              gp(2), gp(3), gp(4), gp(5), gp(6), gp(7), per, Gn, B, side, side, C, Ch ** -0.5, TC_BF16, st)
        torch.cuda.synchronize()
        t = stats[nst:].cpu().numpy().view(np.int64).reshape(Bt * 8, 8)
        d = np.diff(t[:, :6], axis=1).astype(np.float64)
        names = ["loads+fill", "grams+dc", "window dX", "window dW", "token loop"]
        import ctypes
        LL = ctypes.CDLL(os.environ["TC_LIB_PATH"])
        buf = np.zeros(Bt * 8 * 4, dtype=np.int64)
        LL.tc_dbg_wg_stamps(ctypes.c_void_p(buf.ctypes.data), Bt * 8 * 4)
        wgs = np.diff(buf.reshape(-1, 4), axis=1)
        print(f"C={C} backward: window dW inner stamps (mean cycles): x loop {wgs[:,0].mean():.0f}  reductions {wgs[:,1].mean():.0f}  atomics {wgs[:,2].mean():.0f}")
        print(f"C={C} backward, per-phase mean cycles by window:")
        for wname, hs in (("3x3", (0, 1)), ("5x5", (2, 3, 4)), ("7x7", (5, 6, 7))):
            m = np.isin(t[:, 7], hs)
            print("    " + wname + "  " + "  ".join(f"{n} {d[m, i].mean():7.0f}" for i, n in enumerate(names)) + f"  total {(t[m, 5] - t[m, 0]).mean():7.0f}")
