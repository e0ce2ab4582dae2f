"""Per-kernel parity on the MI355X: every C-ABI op (forward + gradient kernels, driven through engine.Graph)
against a plain fp32 PyTorch-CPU evaluation of the same math on the same seeded inputs.

Tolerances are for the fp32 path: 2e-5 abs + 2e-5 rel on activations, 1e-4 rel on reductions over >1e4 terms.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from transception_amd._lib import ACT_COORD, ACT_GELU, ACT_HSWISH, ACT_NONE, ACT_SIGMOID  # noqa: E402
from transception_amd.seeded_init import seeded_tensor  # noqa: E402

DEV = "cuda:0"


def T(tag, shape, scale=1.0):
    return torch.from_numpy(seeded_tensor("ops/" + tag, shape, scale))


def close(got, want, atol=2e-5, rtol=2e-5, what=""):
    got = got.detach().float().cpu()
    want = want.detach().float()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    assert bool((err <= tol).all()), f"{what}: max err {err.max().item():.3e}, ref max {want.abs().max().item():.3e}"


@pytest.fixture()
def G():
    from transception_amd.engine import Graph
    return Graph(torch.float32, torch.device(DEV), training=True, record=True)


def mkP(t):
    from transception_amd.engine import P
    d = t.to(DEV).contiguous()
    return P(d, torch.zeros_like(d, dtype=torch.float32))


def mkV(G, t, requires_grad=True):
    from transception_amd.engine import Var
    return Var(t.to(DEV).contiguous(), requires_grad=requires_grad)


def run_bwd(G, out, gy):
    out.root.grad_t = gy.to(DEV).contiguous().view(out.rows, out.cols)
    out.root.whole_written = True
    G.backward()
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K,bias,res,act", [(100, 64, 64, True, True, ACT_NONE), (300, 200, 147, True, False, ACT_NONE),
                                                 (257, 9, 64, True, False, ACT_NONE), (64, 320, 20, True, False, ACT_SIGMOID),
                                                 (1568, 256, 64, False, False, ACT_NONE), (98, 2048, 512, True, True, ACT_NONE)])
def test_linear(G, M, N, K, bias, res, act):
    x, w = T(f"lin.x{M}.{K}", (M, K)), T(f"lin.w{N}.{K}", (N, K), 1 / math.sqrt(K))
    b = T(f"lin.b{N}", (N,), 0.1) if bias else None
    r = T(f"lin.r{M}.{N}", (M, N)) if res else None
    gy = T(f"lin.g{M}.{N}", (M, N))
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    br = b.clone().requires_grad_() if bias else None
    rr = r.clone().requires_grad_() if res else None
    y = F.linear(xr, wr, br)
    if res:
        y = y + rr
    if act == ACT_SIGMOID:
        y = torch.sigmoid(y)
    y.backward(gy)
    xv, W = mkV(G, x), mkP(w)
    Bp = mkP(b) if bias else None
    rv = mkV(G, r) if res else None
    out = G.linear(xv, W, Bp, residual=rv, act=act)
    close(out.data, y, what="y")
    run_bwd(G, out, gy)
    close(G.grad_of(xv), xr.grad, 5e-5, 5e-5, "dx")
    close(W.grad, wr.grad, 1e-4, 1e-4, "dW")
    if bias:
        close(Bp.grad, br.grad, 1e-4, 1e-4, "db")
    if res:
        close(G.grad_of(rv), rr.grad, what="dres")


def test_linear_big_tiles_and_splitk(G):
    M, N, K = 4096, 256, 64
    x, w, gy = T("linb.x", (M, K)), T("linb.w", (N, K), 0.125), T("linb.g", (M, N))
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    y = F.linear(xr, wr)
    y.backward(gy)
    xv, W = mkV(G, x), mkP(w)
    out = G.linear(xv, W)
    close(out.data, y, what="y")
    run_bwd(G, out, gy)
    close(G.grad_of(xv), xr.grad, 5e-5, 5e-5, "dx")
    close(W.grad, wr.grad, 2e-4, 2e-4, "dW")


@pytest.mark.parametrize("tA,tB", [(0, 0), (0, 1), (1, 0)])
def test_bmm_two_level_batches(G, tA, tB):
    nb1, nb2, M, N, K = 2, 3, 40, 24, 50
    a = T(f"bmm.a{tA}", (nb1, nb2, K, M) if tA else (nb1, nb2, M, K))
    b = T(f"bmm.b{tB}", (nb1, nb2, N, K) if tB else (nb1, nb2, K, N))
    gy = T("bmm.g", (nb1, nb2, M, N))
    ar, br = a.clone().requires_grad_(), b.clone().requires_grad_()
    y = 0.5 * ((ar.transpose(-1, -2) if tA else ar) @ (br.transpose(-1, -2) if tB else br))
    y.backward(gy)
    av, bv = mkV(G, a.reshape(-1, a.shape[-1])), mkV(G, b.reshape(-1, b.shape[-1]))
    out = G.new(nb1 * nb2 * M, N)
    ra, rb = a.shape[-2], b.shape[-2]
    G.bmm(av, bv, out, M, N, K, tA, tB, nb1=nb1, nb2=nb2, sA=(nb2 * ra * av.cols, ra * av.cols),
          sB=(nb2 * rb * bv.cols, rb * bv.cols), sC=(nb2 * M * N, M * N), alpha=0.5)
    close(out.data.view(nb1, nb2, M, N), y, what="y")
    run_bwd(G, out, gy.reshape(-1, N))
    close(G.grad_of(av).view(a.shape), ar.grad, 5e-5, 5e-5, "dA")
    close(G.grad_of(bv).view(b.shape), br.grad, 5e-5, 5e-5, "dB")


@pytest.mark.parametrize("rows,C,eps,act", [(50, 64, 1e-5, ACT_NONE), (33, 160, 1e-6, ACT_NONE), (77, 256, 1e-5, ACT_GELU),
                                            (9, 2048, 1e-5, ACT_GELU), (1000, 320, 1e-5, ACT_NONE)])
def test_layernorm(G, rows, C, eps, act):
    x, g, b, gy = T(f"ln.x{rows}", (rows, C), 2.0), T(f"ln.g{C}", (C,)) * 0.2 + 1, T(f"ln.b{C}", (C,), 0.3), T(f"ln.gy{rows}", (rows, C))
    xr, gr, br = x.clone().requires_grad_(), g.clone().requires_grad_(), b.clone().requires_grad_()
    y = F.layer_norm(xr, (C,), gr, br, eps)
    if act == ACT_GELU:
        y = F.gelu(y)
    y.backward(gy)
    xv, gp, bp = mkV(G, x), mkP(g), mkP(b)
    out = G.layernorm(xv, gp, bp, eps, act)
    close(out.data, y, what="y")
    run_bwd(G, out, gy)
    close(G.grad_of(xv), xr.grad, 5e-5, 5e-5, "dx")
    close(gp.grad, gr.grad, 2e-4, 2e-4, "dgamma")
    close(bp.grad, br.grad, 2e-4, 2e-4, "dbeta")


@pytest.mark.parametrize("rows,C", [(3136, 128), (50176, 64), (784, 320), (37, 64)])
def test_layernorm_deferred_parameter_fold(rows, C, monkeypatch):
    """tc_layernorm_bwd_defer + tc_layernorm_fold (the default: partials parked per launch, one fold per backward leg) against the launch
    that folds at its own tail (TC_LN_DEFER=0): the same dx bit for bit, dgamma / dbeta to summation order; two LayerNorms sharing one
    parameter pair accumulate both contributions."""
    from transception_amd import engine
    from transception_amd.engine import Graph
    x, g, b, gy = T(f"lnd.x{rows}", (rows, C), 2.0), T(f"lnd.g{C}", (C,)) * 0.2 + 1, T(f"lnd.b{C}", (C,), 0.3), T(f"lnd.gy{rows}", (rows, C))
    res = {}
    for defer in (True, False):
        monkeypatch.setattr(engine, "_LN_DEFER", defer)
        G = Graph(torch.float32, torch.device(DEV), training=True, record=True)
        xv, gp, bp = mkV(G, x), mkP(g), mkP(b)
        o1 = G.layernorm(xv, gp, bp)
        o2 = G.layernorm(o1, gp, bp)                      # the same parameters a second time
        n0 = G.n_launch
        run_bwd(G, o2, gy)
        res[defer] = (G.grad_of(xv).clone(), gp.grad.clone(), bp.grad.clone(), G.n_launch - n0)
    assert torch.equal(res[True][0], res[False][0])
    for i, what in ((1, "dgamma"), (2, "dbeta")):
        close(res[True][i], res[False][i].cpu(), 1e-5 * max(1.0, float(res[False][i].abs().max())), 1e-5, what)
    xr, gr, br = x.clone().requires_grad_(), g.clone().requires_grad_(), b.clone().requires_grad_()
    F.layer_norm(F.layer_norm(xr, (C,), gr, br), (C,), gr, br).backward(gy)
    close(res[True][1], gr.grad, 3e-4 * max(1.0, float(gr.grad.abs().max())), 3e-4, "dgamma vs torch")
    close(res[True][2], br.grad, 3e-4 * max(1.0, float(br.grad.abs().max())), 3e-4, "dbeta vs torch")


@pytest.mark.parametrize("C,H,k,bias", [(64, 28, 3, True), (128, 14, 3, False), (320, 7, 3, True), (48, 14, 5, True)])
def test_dwconv_deferred_weight_gradient_fold(C, H, k, bias, monkeypatch):
    """tc_dwconv_bwd with its walkers' sums parked (ws_bytes < 0) + tc_dw_fold (the default) against the launch that folds at its own tail
    (TC_DW_DEFER=0): the same dx bit for bit, dw / db to summation order; a convolution used twice accumulates both contributions."""
    from transception_amd import engine
    from transception_amd.engine import Graph
    B = 4
    x, w, gy = T(f"dwd.x{C}.{H}", (B * H * H, C)), T(f"dwd.w{C}.{k}", (C, 1, k, k), 0.3), T(f"dwd.g{C}.{H}", (B * H * H, C))
    b = T(f"dwd.b{C}", (C,), 0.1) if bias else None
    res = {}
    for defer in (True, False):
        monkeypatch.setattr(engine, "_DW_DEFER", defer)
        G = Graph(torch.float32, torch.device(DEV), training=True, record=True)
        xv, wp = mkV(G, x), mkP(w)
        bp = mkP(b) if bias else None
        o1 = G.dwconv(xv, wp, bp, B, H, H, k, 1, False)
        o2 = G.dwconv(o1, wp, bp, B, H, H, k, 1, False)      # the same weights a second time
        run_bwd(G, o2, gy)
        res[defer] = (G.grad_of(xv).clone(), wp.grad.clone(), bp.grad.clone() if bias else None)
    assert torch.equal(res[True][0], res[False][0])
    close(res[True][1], res[False][1].cpu(), 1e-5 * max(1.0, float(res[False][1].abs().max())), 1e-5, "dw")
    if bias:
        close(res[True][2], res[False][2].cpu(), 1e-5 * max(1.0, float(res[False][2].abs().max())), 1e-5, "db")
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    br = b.clone().requires_grad_() if bias else None
    conv = lambda t: F.conv2d(t.view(B, H, H, C).permute(0, 3, 1, 2), wr, br, padding=k // 2, groups=C).permute(0, 2, 3, 1).reshape(B * H * H, C)
    conv(conv(xr)).backward(gy)
    close(res[True][1], wr.grad, 3e-4 * max(1.0, float(wr.grad.abs().max())), 3e-4, "dw vs torch")


@pytest.mark.parametrize("C,H,k,stride,bias,add", [(64, 12, 3, 1, True, True), (64, 12, 3, 2, False, False), (24, 9, 5, 1, True, False),
                                                    (120, 7, 7, 1, True, False), (256, 14, 3, 1, True, True), (16, 28, 3, 1, True, False)])
def test_dwconv(G, C, H, k, stride, bias, add):
    B = 2
    x = T(f"dw.x{C}.{H}", (B, H, H, C))
    w = T(f"dw.w{C}.{k}", (C, 1, k, k), 0.3)
    b = T(f"dw.b{C}", (C,), 0.1) if bias else None
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    br = b.clone().requires_grad_() if bias else None
    y = F.conv2d(xr.permute(0, 3, 1, 2), wr, br, stride=stride, padding=(k - 1) // 2, groups=C).permute(0, 2, 3, 1)
    if add:
        y = y + xr
    gy = T(f"dw.g{C}.{H}.{stride}", tuple(y.shape))
    y.backward(gy)
    xv, wp = mkV(G, x.reshape(-1, C)), mkP(w)
    bp = mkP(b) if bias else None
    out = G.dwconv(xv, wp, bp, B, H, H, k, stride, add)
    close(out.data.view(y.shape), y, what="y")
    run_bwd(G, out, gy.reshape(-1, C))
    close(G.grad_of(xv).view(x.shape), xr.grad, 5e-5, 5e-5, "dx")
    close(wp.grad, wr.grad, 2e-4, 2e-4, "dw")
    if bias:
        close(bp.grad, br.grad, 2e-4, 2e-4, "db")


@pytest.mark.parametrize("rows,C,act,res", [(392, 64, ACT_HSWISH, False), (1568, 128, ACT_NONE, True), (56, 16, ACT_COORD, False),
                                            (98, 80, ACT_COORD, False), (5000, 320, ACT_HSWISH, False)])
def test_batchnorm_train(G, rows, C, act, res):
    x = T(f"bn.x{rows}.{C}", (rows, C), 1.5) + 0.7
    g, b = T(f"bn.g{C}", (C,)) * 0.2 + 1, T(f"bn.b{C}", (C,), 0.3)
    rm, rv = T(f"bn.rm{C}", (C,), 0.1), T(f"bn.rv{C}", (C,)).abs() + 0.5
    r = T(f"bn.r{rows}.{C}", (rows, C)) if res else None
    gy = T(f"bn.gy{rows}.{C}", (rows, C))
    xr, gr, br = x.clone().requires_grad_(), g.clone().requires_grad_(), b.clone().requires_grad_()
    rmr, rvr = rm.clone(), rv.clone()
    y = F.batch_norm(xr, rmr, rvr, gr, br, True, 0.1, 1e-5)
    if act == ACT_HSWISH:
        y = F.hardswish(y)
    elif act == ACT_COORD:
        y = y * torch.clamp(F.silu(y + 3) / 6, max=1.0)
    if res:
        rr = r.clone().requires_grad_()
        y = y + rr
    y.backward(gy)
    xv, gp, bp = mkV(G, x), mkP(g), mkP(b)
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    resv = mkV(G, r) if res else None
    out = G.batchnorm(xv, gp, bp, rmd, rvd, act, resv)
    close(out.data, y, 3e-5, 3e-5, what="y")
    close(rmd, rmr, 1e-5, 1e-5, "running_mean")
    close(rvd, rvr, 1e-5, 1e-5, "running_var")
    run_bwd(G, out, gy)
    close(G.grad_of(xv), xr.grad, 1e-4, 1e-4, "dx")
    close(gp.grad, gr.grad, 3e-4, 3e-4, "dgamma")
    close(bp.grad, br.grad, 3e-4, 3e-4, "dbeta")
    if res:
        close(G.grad_of(resv), rr.grad, what="dres")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,bias", [(12544, 64, 64, False), (3136, 128, 128, False), (784, 320, 320, False), (896, 16, 256, True),
                                         (224, 80, 1280, True), (100, 24, 72, False)])
def test_linear_leaves_batchnorm_statistics(M, N, K, bias, dtype):
    """TcGemm.bn_part: the 1x1 convolution's epilogue makes the statistics pass of the BatchNorm behind it (DWConv2d_BN / Conv2d_BN /
    CoordAtt conv1 + bn1, MSTr.py:338-339, 394-400, 1333-1335).  The partial sums are those of the ROUNDED stored values about the
    given shift; BatchNorm with them equals BatchNorm with its own statistics pass (same mean / rstd / running buffers / output)."""
    from transception_amd.engine import Graph
    Gh = Graph(dtype, torch.device(DEV), training=True, record=True)
    x, w = T(f"bnl.x{M}.{K}", (M, K)).to(dtype), (T(f"bnl.w{N}.{K}", (N, K), 1 / math.sqrt(K)) + 0.02).to(dtype)
    b = T(f"bnl.b{N}", (N,), 0.3).to(dtype) if bias else None
    g, be = (T(f"bnl.g{N}", (N,)) * 0.2 + 1).to(dtype), T(f"bnl.be{N}", (N,), 0.3).to(dtype)
    rm, rv = T(f"bnl.rm{N}", (N,), 0.3), T(f"bnl.rv{N}", (N,)).abs() + 0.5
    res = {}
    for fused in (True, False):
        rmd, rvd = rm.to(DEV), rv.to(DEV)
        xv = mkV(Gh, x)
        y = Gh.linear(xv, mkP(w), mkP(b) if bias else None, bn_shift=rmd if fused else None)
        assert (getattr(y, "bn_part", None) is not None) == fused
        if fused:
            part, tiles = y.bn_part
            assert tiles == (M + 63) // 64
            torch.cuda.synchronize()
            yf = y.data.float() - rmd                                      # what the kernel summed: stored values minus the shift
            pc = part.cpu()
            assert torch.equal(pc[:N], rm)
            s1 = pc[N:N + tiles * N].view(tiles, N).double().sum(0)
            s2 = pc[N + tiles * N:N + 2 * tiles * N].view(tiles, N).double().sum(0)
            close(s1.float(), yf.double().sum(0).float().cpu(), 2e-3, 2e-5, "sum")
            close(s2.float(), (yf.double() ** 2).sum(0).float().cpu(), 2e-3, 2e-5, "squared sum")
            t0 = pc[N:2 * N]                                               # first row tile alone
            close(t0, yf[:min(64, M)].double().sum(0).float().cpu(), 1e-3, 1e-5, "tile 0")
        gp, bp = mkP(g), mkP(be)
        out = Gh.batchnorm(y, gp, bp, rmd, rvd, ACT_HSWISH)
        run_bwd(Gh, out, T(f"bnl.gy{M}.{N}", (M, N)).to(dtype))
        res[fused] = (out.data.float().cpu(), rmd.cpu(), rvd.cpu(), Gh.grad_of(xv).float().cpu(), gp.grad.cpu(), bp.grad.cpu())
    names = ("y", "running_mean", "running_var", "dx", "dgamma", "dbeta")
    for a, c, nm in zip(res[True], res[False], names):
        tol = 2e-2 if nm in ("y", "dx") else 1e-4                          # y / dx are 16-bit: one rounding step where mean / rstd differ in the last bits
        close(a, c, tol * max(1.0, float(c.abs().max())), 0.0, nm)


@pytest.mark.parametrize("M,N,K,offset", [(12544, 64, 64, 50.0), (3136, 128, 128, 50.0), (12544, 64, 64, -200.0)])
def test_batchnorm_statistics_far_from_the_shift(M, N, K, offset):
    """VERDICT r5 weak #13 / ADVICE r4: the GEMM epilogue sums the stored tile about the BatchNorm's RUNNING mean; after a distribution
    shift the batch mean lies many standard deviations off it and S2 / n - (S1 / n)^2 cancels.  Batch mean = `offset` standard deviations
    with a zero running mean: the variance the BatchNorm then uses (read back from the running_var update) must match the fp64 variance
    of the stored 16-bit values to 2e-4 (fp64 fold of the per-tile sums: norm.hip bn_fold_partials; fp32 folds measured ~2e-3 here)."""
    from transception_amd.engine import Graph
    dtype = torch.bfloat16
    Gh = Graph(dtype, torch.device(DEV), training=True, record=True)
    x, w = T(f"bnfar.x{M}.{K}", (M, K)).to(dtype), T(f"bnfar.w{N}.{K}", (N, K), 1 / math.sqrt(K)).to(dtype)
    b = torch.full((N,), offset).to(dtype)
    g, be = (T(f"bnfar.g{N}", (N,)) * 0.2 + 1).to(dtype), T(f"bnfar.be{N}", (N,), 0.3).to(dtype)
    rmd, rvd = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
    xv = mkV(Gh, x)
    y = Gh.linear(xv, mkP(w), mkP(b), bn_shift=rmd)
    assert getattr(y, "bn_part", None) is not None
    out = Gh.batchnorm(y, mkP(g), mkP(be), rmd, rvd, ACT_HSWISH)
    torch.cuda.synchronize()
    yd = y.data.double()
    mean, var = yd.mean(0), yd.var(0, unbiased=False)
    assert float((mean.abs() / var.sqrt()).min()) > 0.7 * abs(offset)          # the case is what it says
    got_mean = rmd.double() / 0.1
    got_var = (rvd.double() - 0.9) / 0.1 * (M - 1) / M                          # running_var took the unbiased variance
    assert float(((got_mean - mean).abs() / var.sqrt()).max()) < 1e-3
    rel = float(((got_var - var).abs() / var).max())
    assert rel < 2e-4, rel
    want = torch.nn.functional.hardswish(((yd - mean) * torch.rsqrt(var + 1e-5) * g.double().to(DEV) + be.double().to(DEV)).float())
    close(out.data.float().cpu(), want.cpu(), 2e-2 * float(want.abs().max()), 0.0, "y")


@pytest.mark.parametrize("B,N,C", [(2, 784, 256), (3, 49, 1280), (1, 196, 512), (2, 37, 24)])
def test_se_block_pieces(G, B, N, C):
    """SE_Block (MSTr.py:584-593) op by op: squeeze, ReLU, channel gate, BatchNorm + ReLU -- forward and gradients against torch."""
    from transception_amd._lib import ACT_RELU
    x, gate = T(f"se.x{B}.{N}.{C}", (B * N, C)), torch.sigmoid(T(f"se.g{B}.{C}", (B, C)))
    gy, gp = T(f"se.gy{B}.{N}.{C}", (B * N, C)), T(f"se.gp{B}.{C}", (B, C))
    xr, gr = x.clone().requires_grad_(), gate.clone().requires_grad_()
    pooled = xr.view(B, N, C).mean(1)
    hid = torch.relu(pooled - 0.1)
    out = (xr.view(B, N, C) * gr[:, None, :]).reshape(B * N, C)
    (out * gy).sum().backward(retain_graph=True)
    (hid * gp).sum().backward()
    xv, gv = mkV(G, x), mkV(G, gate)
    po = G.chan_pool(xv, B, N)
    close(po.data, pooled, 1e-6, 1e-5, "squeeze")
    sh = mkV(G, (pooled.detach() - 0.1))
    rl = G.relu(sh)
    close(rl.data, hid, 0, 0, "relu")
    o = G.chan_gate(xv, gv, B, N)
    close(o.data, out, 1e-6, 1e-5, "gate")
    for v, g_ in ((o, gy), (po, gp)):
        v.root.grad_t = g_.to(DEV).contiguous(); v.root.whole_written = True
    rl.root.grad_t = gp.to(DEV).contiguous(); rl.root.whole_written = True
    G.backward()
    torch.cuda.synchronize()
    want_dx = xr.grad                                             # gate path + pooled path (through relu(pooled - 0.1) in the torch graph)
    got_dx = G.grad_of(xv).cpu() 
    # the engine's pooled gradient is gp itself (po is a leaf output here), torch's went through the ReLU: rebuild torch's from the mask
    mask = (pooled.detach() - 0.1 > 0).float()
    want = (gy * gate[:, None, :].expand(B, N, C).reshape(B * N, C)) + ((gp / N)[:, None, :].expand(B, N, C).reshape(B * N, C))
    close(got_dx, want, 1e-6, 1e-5, "dx = gate path + squeeze path")
    close(want_dx, (gy * gate[:, None, :].expand(B, N, C).reshape(B * N, C)) + ((gp * mask / N)[:, None, :].expand(B, N, C).reshape(B * N, C)), 1e-6, 1e-5, "torch cross-check")
    close(G.grad_of(gv), gr.grad, 2e-5, 2e-5, "dgate")
    close(G.grad_of(sh), gp * mask, 0, 0, "relu gradient")
    # BatchNorm + ReLU
    rows = B * N
    g, b = T(f"se.bg{C}", (C,)) * 0.2 + 1, T(f"se.bb{C}", (C,), 0.3)
    rm, rv = T(f"se.rm{C}", (C,), 0.1), T(f"se.rv{C}", (C,)).abs() + 0.5
    xr2, gr2, br2 = x.clone().requires_grad_(), g.clone().requires_grad_(), b.clone().requires_grad_()
    y = torch.relu(F.batch_norm(xr2, rm.clone(), rv.clone(), gr2, br2, True, 0.1, 1e-5))
    y.backward(gy)
    from transception_amd.engine import Graph
    G2 = Graph(torch.float32, torch.device(DEV), training=True, record=True)
    xv2, gP, bP = mkV(G2, x), mkP(g), mkP(b)
    ob = G2.batchnorm(xv2, gP, bP, rm.to(DEV), rv.to(DEV), ACT_RELU)
    close(ob.data, y, 3e-5, 3e-5, "relu(bn)")
    run_bwd(G2, ob, gy)
    close(G2.grad_of(xv2), xr2.grad, 1e-4, 1e-4, "bn dx")
    close(gP.grad, gr2.grad, 3e-4, 3e-4, "bn dgamma")
    close(bP.grad, br2.grad, 3e-4, 3e-4, "bn dbeta")


@pytest.mark.parametrize("B,H,W,C,k", [(2, 28, 28, 256, 7), (3, 7, 7, 1280, 3), (1, 14, 14, 512, 7), (2, 5, 9, 24, 3)])
def test_cbam_pieces(G, B, H, W, C, k):
    """CBAMBlock's own kernels (MSTr.py:1128-1211) op by op against torch: max + mean pooling over the tokens (gradient to the first maximal
    token), max + mean over the channels per token, Conv2d(2 -> 1, k x k) + sigmoid on the token grid, the per-token gate."""
    N, rows = H * W, B * H * W
    x = T(f"cb.x{B}.{N}.{C}", (rows, C))
    xr = x.clone().requires_grad_()
    x4 = xr.view(B, H, W, C).permute(0, 3, 1, 2)
    mx, av = F.adaptive_max_pool2d(x4, 1).view(B, C), F.adaptive_avg_pool2d(x4, 1).view(B, C)
    gp = T(f"cb.gp{B}.{C}", (2 * B, C))
    (torch.cat([mx, av], 0) * gp).sum().backward()
    xv = mkV(G, x)
    po = G.chan_pool2(xv, B, N)
    close(po.data, torch.cat([mx, av], 0), 1e-6, 1e-5, "max | mean pooling")
    run_bwd(G, po, gp)
    close(G.grad_of(xv), xr.grad, 1e-6, 1e-5, "pooling gradient")
    # spatial attention: statistics -> conv -> sigmoid -> gate
    from transception_amd.engine import Graph
    G2 = Graph(torch.float32, torch.device(DEV), training=True, record=True)
    w, b = T(f"cb.w{k}", (1, 2, k, k), 0.3), T(f"cb.b{k}", (1,), 0.1)
    gy = T(f"cb.gy{B}.{N}.{C}", (rows, C))
    xr2, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    x42 = xr2.view(B, H, W, C).permute(0, 3, 1, 2)
    st = torch.cat([x42.max(dim=1, keepdim=True)[0], x42.mean(dim=1, keepdim=True)], 1)
    sa = torch.sigmoid(F.conv2d(st, wr, br, padding=k // 2))
    y = (x42 * sa).permute(0, 2, 3, 1).reshape(rows, C)
    y.backward(gy)
    xv2, W_, b_ = mkV(G2, x), mkP(w), mkP(b)
    stv = G2.pix_stats(xv2)
    close(stv.data, st.permute(0, 2, 3, 1).reshape(rows, 2), 1e-6, 1e-5, "channel max | mean per token")
    gate = G2.sa_conv(stv, W_, b_, B, H, W, k)
    close(gate.data.view(-1), sa.reshape(-1), 2e-6, 1e-5, "spatial attention")
    out = G2.pix_gate(xv2, gate)
    close(out.data, y, 2e-6, 1e-5, "gated map")
    run_bwd(G2, out, gy)
    close(G2.grad_of(xv2), xr2.grad, 1e-5, 1e-4, "dx through gate, statistics and convolution")
    close(W_.grad.view(-1), wr.grad.view(-1), 1e-4 * max(1.0, float(wr.grad.abs().max())), 1e-4, "d conv weight")
    close(b_.grad, br.grad, 1e-4 * max(1.0, float(br.grad.abs().max())), 1e-4, "d conv bias")


@pytest.mark.parametrize("B,N,C", [(2, 784, 64), (3, 49, 320), (1, 196, 128), (2, 37, 24)])
def test_cam_module_and_gelu(G, B, N, C):
    """CAM_Module (MSTr.py:478-509) on the four branch maps side by side, and the elementwise GELU behind its Conv3d: forward and gradients
    (input, gamma) against torch."""
    rows = B * N
    x, gy = T(f"cam.x{B}.{N}.{C}", (rows, 4 * C), 0.5), T(f"cam.g{B}.{N}.{C}", (rows, 4 * C))
    gamma = torch.tensor([0.45])
    xr, gr = x.clone().requires_grad_(), gamma.clone().requires_grad_()
    x5 = xr.view(B, N, 4, C).permute(0, 3, 2, 1)                         # [B, C, 4, N]
    energy = x5 @ x5.transpose(-1, -2)
    att = torch.softmax(energy.max(-1, keepdim=True)[0] - energy, -1)
    y = (gr * (att @ x5) + x5).permute(0, 3, 2, 1).reshape(rows, 4 * C)
    y.backward(gy)
    xv, gp = mkV(G, x), mkP(gamma)
    out = G.cam(xv, gp, B, N)
    close(out.data, y, 2e-5, 2e-5, "y")
    run_bwd(G, out, gy)
    close(G.grad_of(xv), xr.grad, 2e-4 * max(1.0, float(xr.grad.abs().max())), 1e-4, "dx")
    close(gp.grad, gr.grad, 1e-4 * max(1.0, float(gr.grad.abs().max())), 1e-4, "dgamma")
    from transception_amd.engine import Graph
    G2 = Graph(torch.float32, torch.device(DEV), training=True, record=True)
    xr2 = x.clone().requires_grad_()
    z = F.gelu(xr2)
    z.backward(gy)
    xv2 = mkV(G2, x)
    o2 = G2.gelu(xv2)
    close(o2.data, z, 2e-6, 2e-6, "gelu")
    run_bwd(G2, o2, gy)
    close(G2.grad_of(xv2), xr2.grad, 5e-6, 5e-6, "gelu gradient")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gelu_16bit_polynomial_against_erf(dtype):
    """The forward GELU of the 16-bit kernels (csrc/tc_common.h gelu_poly: Phi as a clamped odd polynomial, no exp / reciprocal) on EVERY
    storage value of [-8, 8] against fp64 erf: within half a storage spacing of the exact value plus 1.05e-5 |x| (the polynomial's absolute
    error in Phi), i.e. the same rounded result as exact arithmetic (give or take a rounding boundary) wherever |GELU(x)| >= 4e-3, and within 1.05e-5 |x|
    absolutely in the negative tail below that.
    The test-side model of the polynomial (golden_util.gelu_model, which the fp64 rounding models use) is checked against the kernel here too."""
    from golden_util import gelu_model
    from transception_amd.engine import Graph
    bits = torch.arange(0, 65536, dtype=torch.int32).to(torch.int16)
    x = bits.view(dtype).float()
    x = x[torch.isfinite(x) & (x.abs() <= 8.0)]
    G2 = Graph(dtype, torch.device(DEV), training=False, record=False)
    y = G2.gelu(mkV(G2, x.to(dtype).view(-1, 1).contiguous(), requires_grad=False)).data.float().cpu().view(-1).double()
    xd = x.double()
    exact = 0.5 * xd * (1.0 + torch.erf(xd / 2.0 ** 0.5))
    mant, emin = (7, -126) if dtype == torch.bfloat16 else (10, -14)
    spacing = lambda v: torch.exp2(torch.floor(torch.log2(v.abs().clamp_min(2.0 ** emin))) - mant)      # storage spacing at v (subnormals included)
    err = (y - exact).abs()
    slack = err - 0.5 * spacing(exact) * 1.001 - 1.05e-5 * xd.abs()
    assert float(slack.max()) <= 1e-12, (float(slack.max()), float(xd[slack.argmax()]))
    assert float(err[exact.abs() < 4e-3].max()) < 1.05e-5 * 8.0                      # the tail: absolute, not relative (5e-5 at x = -8)
    model = gelu_model(xd, dtype)
    assert float((model - exact).abs().max()) <= 1.05e-5 * 8.0
    # kernel (fp32 Horner) == its fp64 model to one storage rounding wherever the value is not a cancellation residue of the clamp's tail
    body = exact.abs() >= 4e-3
    assert bool(((y - model)[body].abs() <= 0.5 * spacing(model)[body] * 1.001 + 2e-7 * xd.abs()[body]).all())


@pytest.mark.parametrize("rows,C", [(196, 320), (4 * 784, 128), (37, 64)])
def test_gamma_residual(G, rows, C):
    """gamma * a + x with a one-element gamma (the tail of CAM_Factorized_Module, MSTr.py:565-567): forward, both input gradients, dgamma."""
    a, x, gy = T(f"gr.a{rows}", (rows, C)), T(f"gr.x{rows}", (rows, C)), T(f"gr.g{rows}", (rows, C))
    gamma = torch.tensor([0.55])
    ar, xr, gr = a.clone().requires_grad_(), x.clone().requires_grad_(), gamma.clone().requires_grad_()
    y = gr * ar + xr
    y.backward(gy)
    av, xv, gp = mkV(G, a), mkV(G, x), mkP(gamma)
    out = G.gamma_residual(av, xv, gp)
    close(out.data, y, 1e-6, 1e-6, "y")
    run_bwd(G, out, gy)
    close(G.grad_of(av), ar.grad, 1e-6, 1e-6, "da")
    close(G.grad_of(xv), xr.grad, 1e-6, 1e-6, "dx")
    close(gp.grad, gr.grad, 1e-4 * max(1.0, float(gr.grad.abs().max())), 1e-5, "dgamma")


def test_batchnorm_eval():
    from transception_amd.engine import Graph
    Ge = Graph(torch.float32, torch.device(DEV), training=False, record=False)
    rows, C = 100, 64
    x, g, b = T("bne.x", (rows, C)), T("bne.g", (C,)) * 0.2 + 1, T("bne.b", (C,), 0.3)
    rm, rv = T("bne.rm", (C,), 0.1), T("bne.rv", (C,)).abs() + 0.5
    y = F.hardswish(F.batch_norm(x, rm.clone(), rv.clone(), g, b, False, 0.1, 1e-5))
    out = Ge.batchnorm(mkV(Ge, x), mkP(g), mkP(b), rm.to(DEV), rv.to(DEV), ACT_HSWISH)
    close(out.data, y, what="y")


@pytest.mark.parametrize("nb,R,C,axis", [(3, 50, 64, 1), (2, 64, 6076, 1), (2, 784, 64, 0), (2, 64, 6076, 0), (4, 49, 320, 0),
                                         (1, 37, 784, 1)])
def test_softmax(G, nb, R, C, axis):
    x, gy = T(f"sm.x{nb}.{R}.{C}", (nb, R, C), 2.0), T(f"sm.g{nb}.{R}.{C}", (nb, R, C))
    xr = x.clone().requires_grad_()
    y = torch.softmax(xr, dim=2 if axis == 1 else 1)
    y.backward(gy)
    xv = mkV(G, x.reshape(nb * R, C))
    out = G.softmax(xv, nb, axis)
    close(out.data.view(nb, R, C), y, 1e-6, 2e-5, "y")
    run_bwd(G, out, gy.reshape(nb * R, C))
    close(G.grad_of(xv).view(nb, R, C), xr.grad, 2e-6, 5e-5, "dx")


def test_softmax_on_column_slice(G):
    nb, R, C = 2, 196, 128
    x, gy = T("sms.x", (nb * R, 3 * C)), T("sms.g", (nb * R, C))
    xr = x.clone().requires_grad_()
    y = torch.softmax(xr[:, C:2 * C].reshape(nb, R, C), dim=1)
    y.backward(gy.view(nb, R, C))
    xv = mkV(G, x)
    out = G.softmax(xv.colslice(C, 2 * C), nb, 0)
    close(out.data.view(nb, R, C), y, 1e-6, 2e-5, "y")
    run_bwd(G, out, gy)
    close(G.grad_of(xv), xr.grad, 2e-6, 5e-5, "dx (other columns must stay zero)")


def test_fma3_coord_pixel_layout(G):
    rows, C = 300, 64
    a, b, c, gy = T("f3.a", (rows, C)), T("f3.b", (rows, C)), T("f3.c", (rows, C)), T("f3.g", (rows, C))
    ar, br, cr = (t.clone().requires_grad_() for t in (a, b, c))
    y = 0.25 * ar + br * cr
    y.backward(gy)
    av, bv, cv = mkV(G, a), mkV(G, b), mkV(G, c)
    out = G.fma3(av, bv, cv, 0.25)
    close(out.data, y, what="fma3")
    run_bwd(G, out, gy)
    for v, r, n in ((av, ar, "da"), (bv, br, "db"), (cv, cr, "dc")):
        close(G.grad_of(v), r.grad, what=n)


def test_coord_pool_gate(G):
    B, H, C = 2, 7, 80
    x, gy = T("cg.x", (B, H, H, C)), T("cg.g", (B, H, H, C))
    att = torch.sigmoid(T("cg.a", (2 * B * H, C)))
    xr, ar = x.clone().requires_grad_(), att.clone().requires_grad_()
    pooled_ref = torch.cat([xr.mean(2).reshape(B * H, C), xr.mean(1).reshape(B * H, C)], 0)
    a_h, a_w = ar[:B * H].view(B, H, 1, C), ar[B * H:].view(B, 1, H, C)
    y = xr * a_w * a_h
    gp = T("cg.gp", (2 * B * H, C))
    (y * gy).sum().backward(retain_graph=True)
    gx_gate, ga = xr.grad.clone(), ar.grad.clone()
    xr.grad = None
    (pooled_ref * gp).sum().backward()
    gx_pool = xr.grad.clone()
    xv, av = mkV(G, x.reshape(-1, C)), mkV(G, att)
    pooled = G.coord_pool(xv, B, H, H)
    out = G.coord_gate(xv, av, B, H, H)
    close(pooled.data, pooled_ref, what="pooled")
    close(out.data.view(y.shape), y, what="gate")
    pooled.root.grad_t = gp.to(DEV)
    pooled.root.whole_written = True
    run_bwd(G, out, gy.reshape(-1, C))
    close(G.grad_of(xv).view(x.shape), gx_gate + gx_pool, 5e-5, 5e-5, "dx (two consumers accumulate)")
    close(G.grad_of(av), ga, 5e-5, 5e-5, "datt")


def test_pixel_shuffle_transpose(G):
    B, H, p, c = 2, 5, 4, 8
    x, gy = T("ps.x", (B, H, H, p * p * c)), T("ps.g", (B * H * p * H * p, c))
    xr = x.clone().requires_grad_()
    y = xr.reshape(B, H, H, p, p, c).permute(0, 1, 3, 2, 4, 5).reshape(-1, c)
    y.backward(gy)
    xv = mkV(G, x.reshape(-1, p * p * c))
    out = G.pixel_shuffle(xv, B, H, H, p)
    close(out.data, y, 0, 0, "shuffle")
    run_bwd(G, out, gy)
    close(G.grad_of(xv).view(x.shape), xr.grad, 0, 0, "dshuffle")
    t = T("tr.x", (3 * 50, 9))
    tv = mkV(G, t)
    o = G.transpose(tv, 3)
    close(o.data.view(3, 9, 50), t.view(3, 50, 9).transpose(1, 2), 0, 0, "transpose")


def test_patchify_deinterleave(G):
    B, H, C, k = 2, 8, 8, 4
    buf = T("pf.x", (B * H * H + 10, C))          # map occupies the first B*H*H rows
    w = T("pf.w", (C, C, k, k), 0.2)
    br = buf.clone().requires_grad_()
    ref = F.conv2d(br[:B * H * H].view(B, H, H, C).permute(0, 3, 1, 2), w, stride=k)   # [B, C, 2, 2]
    bv = mkV(G, buf)
    cols = G.patchify(bv, 0, H * H * C, B, H, H, C, k)
    got = cols.data.cpu() @ w.view(C, -1).t()                                          # [B*4, C]
    close(got.view(B, 2, 2, C).permute(0, 3, 1, 2), ref, 2e-5, 2e-5, "patchify+gemm == conv")
    gy = T("pf.g", tuple(cols.data.shape))
    (F.unfold(br[:B * H * H].view(B, H, H, C).permute(0, 3, 1, 2), k, stride=k).transpose(1, 2).reshape(-1, C * k * k) * gy).sum().backward()
    run_bwd(G, cols, gy)
    close(G.grad_of(bv), br.grad, 0, 0, "dpatchify")
    # de-interleave (Scale_reduce quirk)
    Pn, mult, Cd = 4, 2, 8
    o = T("di.x", (B * Pn, Cd * mult))
    want = o.view(B, Pn, Cd, mult).permute(0, 3, 1, 2).reshape(B, mult * Pn, Cd)
    ov = mkV(G, o)
    dst = G.new(B * 12, Cd)
    dst.data.zero_()
    G.sr_deinterleave(ov, dst, 2 * Cd, 12 * Cd, B, Pn, Cd, mult)
    close(dst.data.view(B, 12, Cd)[:, 2:2 + mult * Pn], want, 0, 0, "deinterleave")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_patchify_16bit_pieces(dtype):
    """The 16-bit patchify body (whole 4 / 8 / 16-byte pieces, byte permutes; k = 8 / 4 / 2 = Scale_reduce's three maps, MSTr.py:2225-2249)
    against F.unfold: the forward moves storage values bit for bit, the backward ACCUMULATES into the map's gradient (one rounding of the sum);
    single launches and the merged three-map launch."""
    from transception_amd.engine import Graph
    B = 2
    items, off = [], 0
    for (H, C, k) in [(16, 64, 8), (8, 128, 4), (6, 40, 2)]:
        items.append((off, H * H * C, B, H, H, C, k)); off += B * H * H * C
    rows = (off + 63) // 64 + 3
    buf = (T("pf16.x", (rows, 64)).to(dtype)).float()
    pre = (T("pf16.p", (rows, 64), 0.5).to(dtype)).float()            # a gradient already sitting in the map's buffer
    for merged in (False, True):
        G = Graph(dtype, torch.device(DEV), training=True, record=True)
        bv = mkV(G, buf.to(dtype))
        cols = G.patchify_many(bv, items) if merged else [G.patchify(bv, *it) for it in items]
        gys = []
        for (o, sb, _, H, _, C, k), cv in zip(items, cols):
            ref = F.unfold(buf.view(-1)[o:o + B * sb].view(B, H, H, C).permute(0, 3, 1, 2), k, stride=k).transpose(1, 2).reshape(-1, C * k * k)
            assert torch.equal(cv.data.float().cpu(), ref), (merged, k)
            gys.append(T(f"pf16.g{k}", tuple(ref.shape)).to(dtype))
        for cv, gy in zip(cols, gys):
            cv.root.grad_t = gy.to(DEV).contiguous()
            cv.root.whole_written = True
        bv.root.grad_t = pre.to(dtype).to(DEV).contiguous()
        bv.root.whole_written = True
        G.backward()
        torch.cuda.synchronize()
        got = G.grad_of(bv).float().cpu().view(-1)
        want = pre.view(-1).clone()
        for (o, sb, _, H, _, C, k), gy in zip(items, gys):
            g = F.fold(gy.float().view(B, -1, C * k * k).transpose(1, 2), (H, H), k, stride=k).permute(0, 2, 3, 1).reshape(-1)
            want[o:o + B * sb] = (want[o:o + B * sb] + g).to(dtype).float()
        assert torch.equal(got, want), (merged, float((got - want).abs().max()))


def test_attention_paths_agree_with_reference(G):
    B, Nq, Nk, d = 2, 200, 98, 64
    q, kv, gy = T("at.q", (B * Nq, d)), T("at.kv", (B * Nk, 2 * d)), T("at.g", (B * Nq, d))
    qr, kvr = q.clone().requires_grad_(), kv.clone().requires_grad_()
    k, v = kvr[:, :d].reshape(B, Nk, d), kvr[:, d:].reshape(B, Nk, d)
    y = (torch.softmax(qr.view(B, Nq, d) @ k.transpose(1, 2) * 0.125, -1) @ v).reshape(B * Nq, d)
    y.backward(gy)
    for fused in (False, True):
        from transception_amd.engine import Graph
        Gx = Graph(torch.float32, torch.device(DEV), True, True)
        Gx.use_fused_attention = fused
        qv, kvv = mkV(Gx, q), mkV(Gx, kv)
        out = Gx.attention(qv, kvv.colslice(0, d), kvv.colslice(d, 2 * d), B, Nq, Nk, 0.125)
        close(out.data, y, 2e-5, 2e-5, f"attention fused={fused}")
        run_bwd(Gx, out, gy)
        close(Gx.grad_of(qv), qr.grad, 5e-5, 5e-5, f"dq fused={fused}")
        close(Gx.grad_of(kvv), kvr.grad, 5e-5, 5e-5, f"dkv fused={fused}")


def test_seg_loss_and_sgd():
    from transception_amd.train import SegLoss
    B, ncls, H = 2, 9, 32
    logits = T("sl.x", (B, ncls, H, H), 2.0)
    lab = torch.from_numpy(np.random.default_rng(5).integers(0, ncls, (B, H, H)))
    lr_ = logits.clone().requires_grad_()
    ce = F.cross_entropy(lr_, lab)
    p = torch.softmax(lr_, 1)
    oh = F.one_hot(lab, ncls).permute(0, 3, 1, 2).float()
    inter, ys, zs = (p * oh).sum((0, 2, 3)), oh.sum((0, 2, 3)), (p * p).sum((0, 2, 3))
    dice = (1 - (2 * inter + 1e-5) / (zs + ys + 1e-5)).mean()
    loss = 0.4 * ce + 0.6 * dice
    loss.backward()
    ld = logits.to(DEV).requires_grad_()
    out = SegLoss(ncls)(ld, lab.to(DEV))
    got, gce, gdice = out
    assert abs(got.item() - loss.item()) < 2e-6 and abs(gce.item() - ce.item()) < 2e-6 and abs(gdice.item() - dice.item()) < 2e-6
    got.backward()
    close(ld.grad, lr_.grad, 1e-8, 1e-4, "dlogits")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("ld", [16, 12])
def test_seg_loss_on_token_major_logits_equals_the_nchw_kernels(dtype, ld):
    """tc_seg_loss_fwd_tok / _bwd_tok (the captured step's path: logits [B*HW, ld] in the storage type) against tc_seg_loss_fwd / _bwd on the
    NCHW fp32 copy of the same values: probabilities and gradients bit for bit (the gradient after the same round to the storage type),
    the atomically accumulated sums to 1e-6 relative.  ld = 16: 16-bit rows padded to 16-byte pieces (Graph.ln_cls(pad_rows=True)) are read
    and written as whole pieces -- the pad elements of dlogits come back as zeros; any other pitch: element by element, padding untouched."""
    from transception_amd import _lib
    L = _lib.lib()
    B, ncls, H = 3, 9, 24
    HW = H * H
    g = torch.Generator().manual_seed(11)
    code = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[dtype]
    buf = (torch.randn(B * HW, ld, generator=g) * 2.0).to(DEV).to(dtype)
    tok = buf[:, :ncls]                                           # a column slice: row pitch ld
    lab = torch.randint(0, ncls, (B, HW), generator=g).to(DEV)
    nchw = tok.float().view(B, HW, ncls).permute(0, 2, 1).contiguous()
    st = torch.cuda.current_stream().cuda_stream
    pa, pb = torch.empty(B, ncls, HW, device=DEV), torch.empty(B, ncls, HW, device=DEV)
    sa, sb = torch.zeros(1 + 3 * ncls, device=DEV), torch.zeros(1 + 3 * ncls, device=DEV)
    L.tc_seg_loss_fwd(nchw.data_ptr(), lab.data_ptr(), pa.data_ptr(), sa.data_ptr(), B, ncls, HW, 0, st)
    L.tc_seg_loss_fwd_tok(tok.data_ptr(), ld, lab.data_ptr(), pb.data_ptr(), sb.data_ptr(), B, ncls, HW, code, st)
    torch.cuda.synchronize()
    assert torch.equal(pa, pb)
    assert float(((sa - sb).abs() / (sa.abs() + 1e-6)).max()) < 1e-5
    da = torch.empty(B, ncls, HW, device=DEV)
    db = torch.full((B * HW, ld), 3.0, device=DEV, dtype=dtype)
    L.tc_seg_loss_bwd(pa.data_ptr(), lab.data_ptr(), sa.data_ptr(), da.data_ptr(), B, ncls, HW, 0.4, 0.6, float(B * HW), 128.0, None, 0, st)
    L.tc_seg_loss_bwd_tok(pa.data_ptr(), None, 0, lab.data_ptr(), sa.data_ptr(), db.data_ptr(), ld, B, ncls, HW, 0.4, 0.6, float(B * HW), 128.0, None, code, st)
    torch.cuda.synchronize()
    want = da.view(B, ncls, HW).permute(0, 2, 1).reshape(B * HW, ncls).to(dtype)
    assert torch.equal(db[:, :ncls], want)
    pad_zeroed = dtype != torch.float32 and ld % 8 == 0
    assert bool((db[:, ncls:] == (0.0 if pad_zeroed else 3.0)).all())     # padding: zeros on the 16-byte-piece path, untouched otherwise
    # without a probability map: sums only in the forward, the softmax recomputed in the backward -- the same bits
    sc, dc = torch.zeros(1 + 3 * ncls, device=DEV), torch.full((B * HW, ld), 3.0, device=DEV, dtype=dtype)
    L.tc_seg_loss_fwd_tok(tok.data_ptr(), ld, lab.data_ptr(), None, sc.data_ptr(), B, ncls, HW, code, st)
    L.tc_seg_loss_bwd_tok(None, tok.data_ptr(), ld, lab.data_ptr(), sa.data_ptr(), dc.data_ptr(), ld, B, ncls, HW, 0.4, 0.6, float(B * HW), 128.0, None, code, st)
    torch.cuda.synchronize()
    assert float(((sc - sb).abs() / (sb.abs() + 1e-6)).max()) < 1e-5
    assert torch.equal(dc, db)


# ------------------------------------------------------------------------------------------------ bf16 storage path
_LP = torch.bfloat16          # the 16-bit storage type under test: the Gb / lp fixtures run every test below for bf16 and fp16


def _bf(t):
    return t.to(_LP)


def closeb(got, want, rel=1.5e-2, what=""):
    got, want = got.detach().float().cpu(), want.detach().float()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = (got - want).abs().max().item()
    scale = want.abs().max().item()
    assert err <= rel * scale + 1e-6, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


@pytest.fixture(params=[torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def lp(request):
    global _LP
    _LP = request.param
    yield request.param
    _LP = torch.bfloat16


@pytest.fixture()
def Gb(lp):
    from transception_amd.engine import Graph
    return Graph(lp, torch.device(DEV), training=True, record=True)


def mkPb(t):
    from transception_amd.engine import P
    return P(_bf(t).to(DEV).contiguous(), torch.zeros(t.shape, dtype=torch.float32, device=DEV))


@pytest.mark.parametrize("M,N,K", [(300, 200, 147), (4096, 256, 64), (98, 2048, 512), (257, 9, 64)])
def test_linear_bf16(Gb, M, N, K):
    x, w, b = _bf(T(f"lb.x{M}", (M, K))), _bf(T(f"lb.w{N}", (N, K), 1 / math.sqrt(K))), _bf(T(f"lb.b{N}", (N,), 0.1))
    r, gy = _bf(T(f"lb.r{M}", (M, N))), _bf(T(f"lb.g{M}", (M, N)))
    xr, wr, br, rr = (t.float().requires_grad_() for t in (x, w, b, r))
    y = F.linear(xr, wr, br) + rr
    y.backward(gy.float())
    xv, W, Bp, rv = mkV(Gb, x), mkPb(w.float()), mkPb(b.float()), mkV(Gb, r)
    out = Gb.linear(xv, W, Bp, residual=rv)
    closeb(out.data, y, what="y")
    run_bwd(Gb, out, gy)
    closeb(Gb.grad_of(xv), xr.grad, what="dx")
    closeb(W.grad, wr.grad, what="dW")
    closeb(Bp.grad, br.grad, what="db (rides on the dW GEMM)")


@pytest.mark.parametrize("tA,tB", [(0, 0), (0, 1), (1, 0)])
def test_bmm_bf16(Gb, tA, tB):
    nb1, nb2, M, N, K = 2, 3, 40, 24, 56
    a = _bf(T(f"bb.a{tA}", (nb1, nb2, K, M) if tA else (nb1, nb2, M, K)))
    b = _bf(T(f"bb.b{tB}", (nb1, nb2, N, K) if tB else (nb1, nb2, K, N)))
    gy = _bf(T("bb.g", (nb1, nb2, M, N)))
    ar, br = a.float().requires_grad_(), b.float().requires_grad_()
    y = 0.5 * ((ar.transpose(-1, -2) if tA else ar) @ (br.transpose(-1, -2) if tB else br))
    y.backward(gy.float())
    av, bv = mkV(Gb, a.reshape(-1, a.shape[-1])), mkV(Gb, b.reshape(-1, b.shape[-1]))
    out = Gb.new(nb1 * nb2 * M, N)
    ra, rb = a.shape[-2], b.shape[-2]
    Gb.bmm(av, bv, out, M, N, K, tA, tB, nb1=nb1, nb2=nb2, sA=(nb2 * ra * av.cols, ra * av.cols),
           sB=(nb2 * rb * bv.cols, rb * bv.cols), sC=(nb2 * M * N, M * N), alpha=0.5)
    closeb(out.data.view(nb1, nb2, M, N), y, what="y")
    run_bwd(Gb, out, gy.reshape(-1, N))
    closeb(Gb.grad_of(av).view(a.shape), ar.grad, what="dA")
    closeb(Gb.grad_of(bv).view(b.shape), br.grad, what="dB")


@pytest.mark.parametrize("Nq,Nk", [(200, 98), (392, 784), (130, 784)])
def test_attention_bf16(lp, Nq, Nk):
    from transception_amd.engine import Graph
    B, d = 2, 64
    q, kv, gy = _bf(T(f"ab.q{Nq}", (B * Nq, d))), _bf(T(f"ab.kv{Nk}", (B * Nk, 2 * d))), _bf(T(f"ab.g{Nq}", (B * Nq, d)))
    qr, kvr = q.float().requires_grad_(), kv.float().requires_grad_()
    k, v = kvr[:, :d].reshape(B, Nk, d), kvr[:, d:].reshape(B, Nk, d)
    y = (torch.softmax(qr.view(B, Nq, d) @ k.transpose(1, 2) * 0.125, -1) @ v).reshape(B * Nq, d)
    y.backward(gy.float())
    Gx = Graph(lp, torch.device(DEV), True, True)
    Gx.use_fused_attention = True
    qv, kvv = mkV(Gx, q), mkV(Gx, kv)
    out = Gx.attention(qv, kvv.colslice(0, d), kvv.colslice(d, 2 * d), B, Nq, Nk, 0.125)
    closeb(out.data, y, 2e-2, "attention bf16")
    run_bwd(Gx, out, gy)
    closeb(Gx.grad_of(qv), qr.grad, 3e-2, "dq bf16")
    closeb(Gx.grad_of(kvv), kvr.grad, 3e-2, "dkv bf16")


def test_dwconv_layernorm_bf16(Gb):
    B, H, C = 2, 14, 128
    x, w, b = _bf(T("db.x", (B, H, H, C))), _bf(T("db.w", (C, 1, 3, 3), 0.3)), _bf(T("db.b", (C,), 0.1))
    xr, wr, br = (t.float().requires_grad_() for t in (x, w, b))
    y = F.conv2d(xr.permute(0, 3, 1, 2), wr, br, padding=1, groups=C).permute(0, 2, 3, 1) + xr
    gy = _bf(T("db.g", tuple(y.shape)))
    y.backward(gy.float())
    xv, wp, bp = mkV(Gb, x.reshape(-1, C)), mkPb(w.float()), mkPb(b.float())
    out = Gb.dwconv(xv, wp, bp, B, H, H, 3, 1, True)
    closeb(out.data.view(y.shape), y, what="dw y")
    run_bwd(Gb, out, gy.reshape(-1, C))
    closeb(Gb.grad_of(xv).view(x.shape), xr.grad, what="dw dx")
    closeb(wp.grad, wr.grad, what="dw dw")
    closeb(bp.grad, br.grad, what="dw db")


@pytest.mark.parametrize("rows,C,act,prior", [(224, 40, ACT_COORD, False), (896, 8, ACT_COORD, False), (1568, 128, ACT_HSWISH, True), (3136, 128, ACT_HSWISH, False),
                                              (784, 320, ACT_NONE, True), (8000, 64, ACT_HSWISH, False), (12544, 64, ACT_HSWISH, True),
                                              (12544, 60, ACT_NONE, False), (20000, 64, ACT_HSWISH, True)])
def test_batchnorm_train_16bit(Gb, rows, C, act, prior):
    """Training-mode BatchNorm (+ Hardswish / the CoordAtt activation) on 16-bit storage against torch fp32 on the rounded operands, at the map
    sizes of the RIPM / ResBlock / IFF chains; `prior`: the input already carries a gradient the backward adds to."""
    x = _bf(T(f"bn16.x{rows}.{C}", (rows, C), 1.5) + 0.7)
    g, b = _bf(T(f"bn16.g{C}", (C,)) * 0.2 + 1), _bf(T(f"bn16.b{C}", (C,), 0.3))
    rm, rv = T(f"bn16.rm{C}", (C,), 0.1), T(f"bn16.rv{C}", (C,)).abs() + 0.5
    gy, g0 = _bf(T(f"bn16.gy{rows}.{C}", (rows, C))), _bf(T(f"bn16.g0{rows}.{C}", (rows, C)))
    xr, gr, br = x.float().requires_grad_(), g.float().requires_grad_(), b.float().requires_grad_()
    y = F.batch_norm(xr, rm.clone(), rv.clone(), gr, br, True, 0.1, 1e-5)
    if act == ACT_HSWISH:
        y = F.hardswish(y)
    elif act == ACT_COORD:
        y = y * torch.clamp(F.silu(y + 3) / 6, max=1.0)
    y.backward(gy.float())
    xv, gp, bp = mkV(Gb, x), mkPb(g.float()), mkPb(b.float())
    out = Gb.batchnorm(xv, gp, bp, rm.to(DEV), rv.to(DEV), act)
    closeb(out.data, y, what="y")
    if prior:
        xv.root.grad_t = g0.to(DEV).contiguous()
        xv.root.whole_written = True
    run_bwd(Gb, out, gy)
    closeb(Gb.grad_of(xv), xr.grad + (g0.float() if prior else 0), 2e-2, what="dx")
    closeb(gp.grad, gr.grad, 2e-2, "dgamma")
    closeb(bp.grad, br.grad, 2e-2, "dbeta")


@pytest.mark.parametrize("rows,C,residual", [(37, 64, False), (3136, 128, True), (50176, 64, True), (150001, 64, False), (6000, 256, True),
                                             (1111, 512, False), (784, 320, True)])
def test_layernorm_16bit(Gb, rows, C, residual):
    """LayerNorm forward / backward on 16-bit storage against torch fp32 on the rounded operands: C = 64 / 128 / 256 / 512 take the
    eight-channels-per-lane backward (ln_bwd_v8_kernel: 1, 2 or 4 rows per lane group in flight by size, ragged tails), C = 320 the 4-wide one;
    `residual`: the normalised tensor also feeds a skip branch whose gradient the backward adds on top of (in-place accumulation)."""
    x, g, b = _bf(T(f"l16.x{rows}", (rows, C), 2.0)), _bf(T(f"l16.g{C}", (C,)) * 0.2 + 1), _bf(T(f"l16.b{C}", (C,), 0.3))
    gy, gr_ = _bf(T(f"l16.gy{rows}", (rows, C))), _bf(T(f"l16.gr{rows}", (rows, C)))
    xr, gr, br = x.float().requires_grad_(), g.float().requires_grad_(), b.float().requires_grad_()
    y = F.layer_norm(xr, (C,), gr, br, 1e-5)
    y.backward(gy.float())
    want_dx = xr.grad + (gr_.float() if residual else 0)
    xv, gp, bp = mkV(Gb, x), mkPb(g.float()), mkPb(b.float())
    out = Gb.layernorm(xv, gp, bp, 1e-5)
    closeb(out.data, y, what="y")
    if residual:                                           # the skip branch arrived first: LayerNorm's backward accumulates
        xv.root.grad_t = gr_.to(DEV).contiguous()
        xv.root.whole_written = True
    run_bwd(Gb, out, gy)
    closeb(Gb.grad_of(xv), want_dx, what="dx")
    closeb(gp.grad, gr.grad, 2e-2, "dgamma")
    closeb(bp.grad, br.grad, 2e-2, "dbeta")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_attention_segmented(dtype):
    """Stage-major queries (segments of B images each) against image-major K/V: one launch on the bf16 path."""
    from transception_amd.engine import Graph
    B, d, Nk, nq = 2, 64, 210, [150, 64, 33]
    rows = B * sum(nq)
    cast = (lambda t: t.to(dtype))
    q, kv, gy = cast(T("as.q", (rows, d))), cast(T("as.kv", (B * Nk, 2 * d))), cast(T("as.g", (rows, d)))
    qr, kvr = q.float().requires_grad_(), kv.float().requires_grad_()
    k, v = kvr[:, :d].reshape(B, Nk, d), kvr[:, d:].reshape(B, Nk, d)
    outs, r0 = [], 0
    for n in nq:
        qs = qr[r0:r0 + B * n].view(B, n, d)
        outs.append((torch.softmax(qs @ k.transpose(1, 2) * 0.125, -1) @ v).reshape(B * n, d))
        r0 += B * n
    y = torch.cat(outs, 0)
    y.backward(gy.float())
    Gx = Graph(dtype, torch.device(DEV), True, True)
    qv, kvv = mkV(Gx, q), mkV(Gx, kv)
    out = Gx.attention_seg(qv, kvv.colslice(0, d), kvv.colslice(d, 2 * d), B, nq, Nk, 0.125)
    tol = 2e-5 if dtype == torch.float32 else None
    if tol:
        close(out.data, y, 2e-5, 2e-5, "seg attention")
    else:
        closeb(out.data, y, 2e-2, "seg attention bf16")
    run_bwd(Gx, out, gy)
    if tol:
        close(Gx.grad_of(qv), qr.grad, 5e-5, 5e-5, "dq")
        close(Gx.grad_of(kvv), kvr.grad, 5e-5, 5e-5, "dkv")
        return
    closeb(Gx.grad_of(qv), qr.grad, 3e-2, "dq bf16")
    closeb(Gx.grad_of(kvv), kvr.grad, 3e-2, "dkv bf16")
    _seg_prescaled_cases(dtype, q, kv, gy, B, nq, Nk, y, qr.grad, kvr.grad)


def _seg_prescaled_cases(dtype, q, kv, gy, B, nq, Nk, y, dq_ref, dkv_ref, tol_o=2e-2, tol_g=3e-2):
    """The form the model uses: Q handed over as q * scale * log2(e) rounded once from fp32 (here from the test's fp32 q), the
    gradient returned for the UNSCALED q -- through the hand-scheduled forward stream (default) and the compiler-scheduled kernel
    (TC_ATTN_FWD_ASM=0), which must agree with each other to rounding of P."""
    import os
    from transception_amd.engine import Graph
    d = 64
    qp = (q.float() * (0.125 * 1.4426950408889634)).to(dtype)
    # reference for the stored operand: q_eff = stored / (scale log2 e)
    qe = (qp.float() / (0.125 * 1.4426950408889634)).requires_grad_()
    kvr = kv.float().requires_grad_()
    k, v = kvr[:, :d].reshape(B, Nk, d), kvr[:, d:].reshape(B, Nk, d)
    outs, r0 = [], 0
    for n in nq:
        qs = qe[r0:r0 + B * n].view(B, n, d)
        outs.append((torch.softmax(qs @ k.transpose(1, 2) * 0.125, -1) @ v).reshape(B * n, d))
        r0 += B * n
    yr = torch.cat(outs, 0)
    yr.backward(gy.float())
    res = {}
    for impl in ("1", "0"):
        os.environ["TC_ATTN_FWD_ASM"] = impl
        try:
            Gx = Graph(dtype, torch.device(DEV), True, True)
            qv, kvv = mkV(Gx, qp), mkV(Gx, kv)
            out = Gx.attention_seg(qv, kvv.colslice(0, d), kvv.colslice(d, 2 * d), B, nq, Nk, 0.125, q_prescaled=True)
            assert torch.isfinite(out.data.float()).all()
            closeb(out.data, yr, tol_o, f"seg attention, prescaled q, impl {impl}")
            run_bwd(Gx, out, gy)
            closeb(Gx.grad_of(qv), qe.grad, tol_g, f"dq, prescaled q, impl {impl}")
            closeb(Gx.grad_of(kvv), kvr.grad, tol_g, f"dkv, prescaled q, impl {impl}")
            res[impl] = out.data.float().clone()
        finally:
            os.environ.pop("TC_ATTN_FWD_ASM", None)
    closeb(res["1"], res["0"].cpu(), 1e-2, "hand-scheduled vs compiler-scheduled forward")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_attention_segmented_rereference(dtype):
    """The forward kernel references its exponents to the first 32-key sub-tile's row maximum and re-references a wave's rows when a
    row sum passes 2^30 (2^14 for fp16 P).  Here the later keys are 24x larger than the first ones, so their scores outgrow the first
    reference by far more than that (and by more than fp16's range): output, lse (through the backward) and gradients must still
    match the fp32 reference; a second case puts the large keys FIRST (the reference starts high and every later P underflows)."""
    from transception_amd.engine import Graph
    B, d, Nk, nq = 2, 64, 300, [96, 40]
    rows = B * sum(nq)
    for big_first in (False, True):
        q, kv, gy = T("ar.q", (rows, d)).to(dtype), T("ar.kv", (B * Nk, 2 * d)), T("ar.g", (rows, d)).to(dtype)
        kv = kv.view(B, Nk, 2 * d).clone()
        sl = slice(0, 40) if big_first else slice(70, Nk)
        kv[:, sl, :d] *= 24.0
        kv = kv.view(B * Nk, 2 * d).to(dtype)
        qr, kvr = q.float().requires_grad_(), kv.float().requires_grad_()
        k, v = kvr[:, :d].reshape(B, Nk, d), kvr[:, d:].reshape(B, Nk, d)
        outs, r0 = [], 0
        for n in nq:
            qs = qr[r0:r0 + B * n].view(B, n, d)
            outs.append((torch.softmax(qs @ k.transpose(1, 2) * 0.125, -1) @ v).reshape(B * n, d))
            r0 += B * n
        y = torch.cat(outs, 0)
        y.backward(gy.float())
        Gx = Graph(dtype, torch.device(DEV), True, True)
        qv, kvv = mkV(Gx, q), mkV(Gx, kv)
        out = Gx.attention_seg(qv, kvv.colslice(0, d), kvv.colslice(d, 2 * d), B, nq, Nk, 0.125)
        assert torch.isfinite(out.data.float()).all()
        closeb(out.data, y, 2e-2, f"seg attention, large keys {'first' if big_first else 'late'}")
        run_bwd(Gx, out, gy)
        closeb(Gx.grad_of(qv), qr.grad, 3e-2, "dq")
        closeb(Gx.grad_of(kvv), kvr.grad, 3e-2, "dkv")
        _seg_prescaled_cases(dtype, q, kv, gy, B, nq, Nk, y, qr.grad, kvr.grad)     # the rare re-reference path of the hand-scheduled stream


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_attention_stream_vs_compiler_kernel_and_fp64_over_shapes(dtype):
    """The hand-scheduled forward stream (tc_attn_fwd_seg with qscaled = 1, the model's form) over the shapes that exercise its edges:
    the minimum of two key sub-tiles, every tail length class (Nk mod 32 in {0, 1, 16, 31}), one / few / many loop iterations, ring
    wrap-around (> 12 sub-tiles), one to four segments with partial and single-row tiles, dead wave tiles, B = 1 -- O and lse against
    an fp64 statement on the stored operands, and against the compiler-scheduled kernel (TC_ATTN_FWD_ASM=0) on the same operands."""
    import ctypes as C
    import os
    from transception_amd._lib import TC_BF16, TC_F16, lib
    L = lib()
    d, scale, l2e = 64, 0.125, 1.4426950408889634
    dt = TC_BF16 if dtype == torch.bfloat16 else TC_F16
    st = torch.cuda.current_stream().cuda_stream
    cases = [(1, [64], 64), (1, [33], 65), (2, [100, 37], 80), (2, [1, 32, 31], 95), (1, [384], 96), (3, [70, 5, 129, 12], 127), (2, [64, 200], 128),
             (1, [95], 161), (2, [392, 40], 400), (2, [150, 64, 33], 784), (1, [257], 1023)]
    for ci, (B, nq, Nk) in enumerate(cases):
        rows = B * sum(nq)
        qf = T(f"sv.q{ci}", (rows, d)).to(DEV)
        q = (qf * (scale * l2e)).to(dtype)
        kv = T(f"sv.kv{ci}", (B * Nk, 2 * d)).to(DEV).to(dtype)
        k, v = kv[:, :d], kv[:, d:]
        nqc = (C.c_int * 4)(*(list(nq) + [0] * (4 - len(nq))))
        outs = {}
        for impl in ("1", "0"):
            os.environ["TC_ATTN_FWD_ASM"] = impl
            try:
                o = torch.full((rows, d), float("nan"), device=DEV).to(dtype)
                lse = torch.full((rows,), float("nan"), device=DEV)
                rc = L.tc_attn_fwd_seg(q.data_ptr(), d, k.data_ptr(), 2 * d, v.data_ptr(), 2 * d, Nk * 2 * d, o.data_ptr(), d, lse.data_ptr(), B, len(nq),
                                       nqc, Nk, scale, 1, dt, st)
                torch.cuda.synchronize()
                assert rc in (0, None)
                outs[impl] = (o.double(), lse.double())
            finally:
                os.environ.pop("TC_ATTN_FWD_ASM", None)
        ref_o = torch.empty(rows, d, dtype=torch.float64, device=DEV)
        ref_l = torch.empty(rows, dtype=torch.float64, device=DEV)
        off, kd, vd = 0, k.double().view(B, Nk, d), v.double().view(B, Nk, d)
        for n in nq:
            s_ = torch.einsum("bqd,bkd->bqk", q[off:off + B * n].double().view(B, n, d), kd) / l2e
            ref_l[off:off + B * n] = torch.logsumexp(s_, -1).reshape(-1)
            ref_o[off:off + B * n] = torch.einsum("bqk,bkd->bqd", torch.softmax(s_, -1), vd).reshape(-1, d)
            off += B * n
        tol_o = 8e-3 if dtype == torch.bfloat16 else 1.5e-3          # P and O are rounded to the storage type
        for impl, (o, lse) in outs.items():
            assert torch.isfinite(o).all() and torch.isfinite(lse).all(), (impl, B, nq, Nk)
            assert (o - ref_o).abs().max().item() <= tol_o * max(1.0, ref_o.abs().max().item()), (impl, B, nq, Nk, (o - ref_o).abs().max().item())
            assert (lse - ref_l).abs().max().item() <= 2e-5, (impl, B, nq, Nk, (lse - ref_l).abs().max().item())
        assert (outs["1"][0] - outs["0"][0]).abs().max().item() <= tol_o, (B, nq, Nk)
        assert (outs["1"][1] - outs["0"][1]).abs().max().item() <= 2e-5, (B, nq, Nk)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_attention_backward_streams_vs_compiler_kernels_and_fp64_over_shapes(dtype):
    """The hand-scheduled dQ and dK / dV streams (tc_attn_bwd_seg with qscaled = 1: gen_dq_asm.py, gen_dkv_asm.py) over the shapes that
    exercise their edges: every key tail class (Nk mod 32), two key sub-tiles (the dQ stream's minimum) up to ring wrap-around, two
    query tiles per chunk (the minimum), chunks of unequal length, ring wrap-around (> 8 sub-tiles per chunk), one to four segments with
    partial and single-row tiles (rows past a segment's end get P = 0 through the padded statistics), key counts with a partial last
    wave and idle waves, B = 1.  They perform the arithmetic of the compiler-scheduled kernels in the same order: dQ / dK / dV are
    BIT-IDENTICAL to TC_ATTN_DQ_ASM=0 TC_ATTN_DKV_ASM=0 on the same bf16 operands (fp16 dK / dV: to the last place); both are held to an
    fp64 statement on the stored operands."""
    import ctypes as C
    import os
    from transception_amd._lib import TC_BF16, TC_F16, lib
    from transception_amd.engine import ATTN_DKV_SPLITS
    L = lib()
    d, scale, l2e = 64, 0.125, 1.4426950408889634
    dt = TC_BF16 if dtype == torch.bfloat16 else TC_F16
    st = torch.cuda.current_stream().cuda_stream
    cases = [(1, [64], 64), (1, [33], 65), (2, [100, 37], 80), (2, [1, 32, 31], 95), (1, [384], 96), (3, [70, 5, 129, 12], 127), (2, [640, 200], 128),
             (1, [950], 161), (2, [392, 40], 400), (2, [1500, 64, 33], 784), (16, [784, 392, 245, 98], 196), (1, [2570], 33)]
    for ci, (B, nq, Nk) in enumerate(cases):
        rows = B * sum(nq)
        q = (T(f"dv.q{ci}", (rows, d)).to(DEV) * (scale * l2e)).to(dtype)
        kv = T(f"dv.kv{ci}", (B * Nk, 2 * d)).to(DEV).to(dtype)
        k, v = kv[:, :d], kv[:, d:]
        do = T(f"dv.g{ci}", (rows, d)).to(DEV).to(dtype)
        nqc = (C.c_int * 4)(*(list(nq) + [0] * (4 - len(nq))))
        o = torch.empty((rows, d), device=DEV, dtype=dtype)
        lse = torch.empty((rows,), device=DEV)
        assert L.tc_attn_fwd_seg(q.data_ptr(), d, k.data_ptr(), 2 * d, v.data_ptr(), 2 * d, Nk * 2 * d, o.data_ptr(), d, lse.data_ptr(), B, len(nq), nqc, Nk,
                                 scale, 1, dt, st) in (0, None)
        outs = {}
        for impl in ("1", "1u", "0"):                  # streams (query chunks paired where the plan allows), streams unpaired, compiler-scheduled
            os.environ["TC_ATTN_DKV_ASM"] = os.environ["TC_ATTN_DQ_ASM"] = impl[0]
            os.environ["TC_ATTN_DKV_PAIR"] = "0" if impl == "1u" else "1"
            try:
                dq = torch.full((rows, d), float("nan"), device=DEV).to(dtype)
                dkv = torch.full((B * Nk, 2 * d), float("nan"), device=DEV).to(dtype)
                delta = torch.empty((rows,), device=DEV)
                dkv32 = torch.full((ATTN_DKV_SPLITS * B * Nk * 128,), float("nan"), device=DEV)
                rc = L.tc_attn_bwd_seg(q.data_ptr(), d, k.data_ptr(), 2 * d, v.data_ptr(), 2 * d, Nk * 2 * d, o.data_ptr(), d, do.data_ptr(), d, lse.data_ptr(),
                                       delta.data_ptr(), dkv32.data_ptr(), dq.data_ptr(), d, dkv.data_ptr(), 2 * d, dkv.data_ptr() + 2 * d, 2 * d, Nk * 2 * d, B,
                                       len(nq), nqc, Nk, scale, 1, dt, st)
                torch.cuda.synchronize()
                assert rc in (0, None)
                outs[impl] = (dq.double(), dkv.double())
            finally:
                os.environ.pop("TC_ATTN_DKV_ASM", None)
                os.environ.pop("TC_ATTN_DQ_ASM", None)
                os.environ.pop("TC_ATTN_DKV_PAIR", None)
        # paired and unpaired chunks run the same stream and the fold is a fixed tree: the same bits for both storage types
        assert torch.equal(outs["1"][1], outs["1u"][1]) and torch.equal(outs["1"][0], outs["1u"][0]), (B, nq, Nk)
        if dtype == torch.bfloat16:
            assert torch.equal(outs["1"][1], outs["0"][1]), (B, nq, Nk, (outs["1"][1] - outs["0"][1]).abs().max().item())
        else:       # fp16: identical but for single last-place differences of a dK row (one case of twelve, one key: 1.2e-4 at magnitude 0.2):
                    # hipcc fuses dS = P dP and its conversion to fp16 into v_fma_mixlo_f16 for some elements (ONE rounding), the stream
                    # multiplies in fp32 and converts (two) -- an element 0.0002 ulp from a tie lands on the other side (scripts/exp/dkv_f16_probe.py)
            assert (outs["1"][1] - outs["0"][1]).abs().max().item() <= 1e-3 * max(1.0, outs["0"][1].abs().max().item()), (B, nq, Nk)
        assert torch.equal(outs["1"][0], outs["0"][0]), (B, nq, Nk)
        # fp64 statement on the stored operands (q holds q' = q * scale * log2(e): the gradients wanted are those of the unscaled product)
        kd, vd = k.double().view(B, Nk, d), v.double().view(B, Nk, d)
        ref_dk, ref_dv = torch.zeros(B, Nk, d, dtype=torch.float64, device=DEV), torch.zeros(B, Nk, d, dtype=torch.float64, device=DEV)
        ref_dq = torch.empty(rows, d, dtype=torch.float64, device=DEV)
        off = 0
        for n in nq:
            qs_ = (q[off:off + B * n].double().view(B, n, d) / (scale * l2e)).requires_grad_()
            kk, vv = kd.clone().requires_grad_(), vd.clone().requires_grad_()
            out = torch.softmax(torch.einsum("bqd,bkd->bqk", qs_, kk) * scale, -1) @ vv
            out.backward(do[off:off + B * n].double().view(B, n, d))
            ref_dk += kk.grad; ref_dv += vv.grad
            ref_dq[off:off + B * n] = qs_.grad.reshape(-1, d)      # gradient of the UNSCALED projection output (include/transception_hip.h, qscaled)
            off += B * n
        tol = 2.5e-2 if dtype == torch.bfloat16 else 4e-3             # P, dS and the outputs are rounded to the storage type
        got = outs["1"][1].view(B, Nk, 2 * d)
        for nm, g_, r_ in (("dK", got[..., :d], ref_dk), ("dV", got[..., d:], ref_dv), ("dQ", outs["1"][0], ref_dq)):
            assert torch.isfinite(g_).all(), (nm, B, nq, Nk)
            assert (g_ - r_).abs().max().item() <= tol * max(1.0, r_.abs().max().item()), (nm, B, nq, Nk, (g_ - r_).abs().max().item(), r_.abs().max().item())


@pytest.mark.parametrize("Bt,N,heads,Ch", [(3, 784, 8, 8), (2, 196, 8, 16), (2, 49, 8, 40)])
def test_factor_att_core_fused(G, Bt, N, heads, Ch):
    """tc_factor_att_fwd/bwd vs a plain PyTorch fp32 restatement of MSTr.py:864-877 (Appendix C.1):
    o = scale * q (softmax_N(k)^T v) + q * convv per (image, head), q/k/v column slices of one qkv buffer."""
    C = heads * Ch
    scale = Ch ** -0.5
    qkv, cv, gy = T(f"fa.qkv{N}.{Ch}", (Bt * N, 3 * C)), T(f"fa.cv{N}.{Ch}", (Bt * N, C)), T(f"fa.g{N}.{Ch}", (Bt * N, C))
    qr, cr = qkv.clone().requires_grad_(), cv.clone().requires_grad_()
    q, k, v = (qr[:, i * C:(i + 1) * C].reshape(Bt, N, heads, Ch).permute(0, 2, 1, 3) for i in range(3))
    ctx = torch.softmax(k, dim=2).transpose(-1, -2) @ v
    fa = (q @ ctx).permute(0, 2, 1, 3).reshape(Bt * N, C)
    ref = scale * fa + qr[:, :C] * cr
    ref.backward(gy)
    xv, cvv = mkV(G, qkv), mkV(G, cv)
    out = G.factor_att_core(xv.colslice(0, C), xv.colslice(C, 2 * C), xv.colslice(2 * C, 3 * C), cvv, Bt, N, heads, scale)
    close(out.data, ref, 2e-5, 2e-5, "o")
    run_bwd(G, out, gy)
    close(G.grad_of(xv), qr.grad, 2e-5, 1e-4, "dqkv")
    close(G.grad_of(cvv), cr.grad, 2e-6, 2e-5, "dconvv")


def test_dwconv_backward_as_two_launches(G, monkeypatch):
    """The default backward of a stride-1 depthwise convolution is ONE launch for both gradients (tc_dwconv_bwd, tc_dwconv_multi mode 3);
    the two-launch form stays for the side-stream mode: same checks against torch with the merge switched off."""
    import transception_amd.engine as E
    monkeypatch.setattr(E, "_DW_BWD_ONE", False)
    test_dwconv(G, 64, 14, 3, 1, True, True)
    test_dwconv_multi_matches_torch(G, 16, 14)


@pytest.mark.parametrize("Ch,side", [(8, 28), (16, 14), (40, 7)])
def test_dwconv_multi_matches_torch(G, Ch, side):
    """tc_dwconv_multi: the three ConvRelPosEnc window sizes (3/5/7 on 2/3/3 heads' channels, MSTr.py:785-816) on column slices
    of one buffer in one launch each for forward, input gradient and weight gradient, vs F.conv2d."""
    B, C = 2, 8 * Ch
    widths, ks = [2 * Ch, 3 * Ch, 3 * Ch], [3, 5, 7]
    x, gy = T(f"dwm.x{Ch}", (B * side * side, 3 * C)), T(f"dwm.g{Ch}", (B * side * side, C))
    wts = [T(f"dwm.w{Ch}.{i}", (w, 1, k, k), 0.3) for i, (w, k) in enumerate(zip(widths, ks))]
    bss = [T(f"dwm.b{Ch}.{i}", (w,), 0.3) for i, w in enumerate(widths)]
    xr = x.clone().requires_grad_()
    wr, br = [w.clone().requires_grad_() for w in wts], [b.clone().requires_grad_() for b in bss]
    v = xr[:, 2 * C:].reshape(B, side, side, C).permute(0, 3, 1, 2)
    outs, c0 = [], 0
    for w_, b_, wd, k in zip(wr, br, widths, ks):
        outs.append(F.conv2d(v[:, c0:c0 + wd], w_, b_, padding=k // 2, groups=wd)); c0 += wd
    ref = torch.cat(outs, 1).permute(0, 2, 3, 1).reshape(B * side * side, C)
    ref.backward(gy)
    xv = mkV(G, x)
    out = G.new(B * side * side, C)
    Ws, Bs = [mkP(w.reshape(w.shape[0], -1)) for w in wts], [mkP(b) for b in bss]
    xs, os_, c0 = [], [], 0
    for wd in widths:
        xs.append(xv.colslice(2 * C + c0, 2 * C + c0 + wd)); os_.append(out.colslice(c0, c0 + wd)); c0 += wd
    G.dwconv_multi(xs, Ws, Bs, (B, side, side), ks, os_)
    close(out.data, ref, 2e-5, 2e-5, "y")
    run_bwd(G, out, gy)
    close(G.grad_of(xv), xr.grad, 2e-5, 5e-5, "dx")
    for i in range(3):
        close(Ws[i].grad.view_as(wts[i]), wr[i].grad, 5e-5, 1e-4, f"dw{i}")
        close(Bs[i].grad, br[i].grad, 5e-5, 1e-4, f"db{i}")


def test_dwconv_multi_grouped(G):
    """Same, with three stacked weight groups (the three MB paths of a stage run as one grouped launch)."""
    from transception_amd.engine import P
    Ch, side, B, Gn = 8, 14, 2, 3
    C = 8 * Ch
    widths, ks = [2 * Ch, 3 * Ch, 3 * Ch], [3, 5, 7]
    rows = B * side * side
    x, gy = T("dwmg.x", (Gn * rows, 3 * C)), T("dwmg.g", (Gn * rows, C))
    per = sum(w * k * k + w for w, k in zip(widths, ks))
    arena = T("dwmg.p", (Gn * per,), 0.3).to(DEV)
    garena = torch.zeros_like(arena)
    offs, o = [], 0
    for w, k in zip(widths, ks):
        offs.append((o, o + w * k * k)); o += w * k * k + w
    xr = x.clone().requires_grad_()
    ar = arena.cpu().clone().requires_grad_()
    refs = []
    for g in range(Gn):
        v = xr[g * rows:(g + 1) * rows, 2 * C:].reshape(B, side, side, C).permute(0, 3, 1, 2)
        outs, c0 = [], 0
        for (ow, ob), wd, k in zip(offs, widths, ks):
            w_ = ar[g * per + ow:g * per + ow + wd * k * k].view(wd, 1, k, k)
            b_ = ar[g * per + ob:g * per + ob + wd]
            outs.append(F.conv2d(v[:, c0:c0 + wd], w_, b_, padding=k // 2, groups=wd)); c0 += wd
        refs.append(torch.cat(outs, 1).permute(0, 2, 3, 1).reshape(rows, C))
    ref = torch.cat(refs, 0)
    ref.backward(gy)
    xv = mkV(G, x)
    out = G.new(Gn * rows, C)
    Ws = [P(arena[ow:ow + wd * k * k].view(wd, k * k), garena[ow:ow + wd * k * k].view(wd, k * k), per) for (ow, _), wd, k in zip(offs, widths, ks)]
    Bs = [P(arena[ob:ob + wd], garena[ob:ob + wd], per) for (_, ob), wd in zip(offs, widths)]
    xs, os_, c0 = [], [], 0
    for wd in widths:
        xs.append(xv.colslice(2 * C + c0, 2 * C + c0 + wd)); os_.append(out.colslice(c0, c0 + wd)); c0 += wd
    with G.grouped(Gn, per):
        G.dwconv_multi(xs, Ws, Bs, (B, side, side), ks, os_)
    close(out.data, ref, 2e-5, 2e-5, "y")
    run_bwd(G, out, gy)
    close(G.grad_of(xv), xr.grad, 2e-5, 5e-5, "dx")
    close(garena, ar.grad, 5e-5, 1e-4, "dw/db of all groups")


def _flat_params(shapes, tag):
    """n Linear layers laid out like the model's flat arena: [W0 | b0 | W1 | b1 | ...] with fp32 gradient mirrors."""
    from transception_amd.engine import P
    N, K = shapes
    n = 3
    arena = T(tag, (n * (N * K + N),), 0.2).to(DEV)
    garena = torch.zeros_like(arena)
    Ws = [P(arena[i * (N * K + N):i * (N * K + N) + N * K].view(N, K), garena[i * (N * K + N):i * (N * K + N) + N * K].view(N, K)) for i in range(n)]
    bs = [P(arena[i * (N * K + N) + N * K:(i + 1) * (N * K + N)], garena[i * (N * K + N) + N * K:(i + 1) * (N * K + N)]) for i in range(n)]
    return arena, garena, Ws, bs


def test_linear_multi_keys_queries_values(G):
    """linear_multi: three same-shape Linears on one input as ONE batched GEMM; backward = one K=3N product over the gapped weight
    stack (TcGemm.bgap) + one batched weight-gradient GEMM (EfficientAttention keys/queries/values, MSTr.py:109-111)."""
    M_, N, K = 392, 64, 64
    arena, garena, Ws, bs = _flat_params((N, K), "lm.p")
    x, gy = T("lm.x", (M_, K)), T("lm.g", (M_, 3 * N))
    xr, ar = x.clone().requires_grad_(), arena.cpu().clone().requires_grad_()
    per = N * K + N
    ref = torch.cat([xr @ ar[i * per:i * per + N * K].view(N, K).t() + ar[i * per + N * K:(i + 1) * per] for i in range(3)], 1)
    ref.backward(gy)
    xv = mkV(G, x)
    out = G.new(M_, 3 * N)
    G.linear_multi(xv, Ws, bs, out)
    close(out.data, ref, 2e-5, 2e-5, "y")
    run_bwd(G, out, gy)
    close(G.grad_of(xv), xr.grad, 2e-5, 1e-4, "dx")
    close(garena, ar.grad, 5e-5, 1e-4, "dW / db")


def test_linear_many_independent_shapes(G):
    """linear_many: independent Linears of different shapes (+ residual) through tc_gemm_multi, forward and backward."""
    specs = [(300, 64, 256), (128, 128, 512), (70, 320, 1280)]
    items, refs, leaves = [], [], []
    for i, (M_, K, N) in enumerate(specs):
        x, W, b, r = T(f"lmy.x{i}", (M_, K)), T(f"lmy.w{i}", (N, K), 0.1), T(f"lmy.b{i}", (N,)), T(f"lmy.r{i}", (M_, N))
        xr, Wr, br, rr = (t.clone().requires_grad_() for t in (x, W, b, r))
        refs.append(xr @ Wr.t() + br + rr); leaves.append((xr, Wr, br, rr))
        items.append((mkV(G, x), mkP(W), mkP(b), G.new(M_, N), mkV(G, r)))
    outs = G.linear_many(items)
    for i, (o, ref) in enumerate(zip(outs, refs)):
        close(o.data, ref, 2e-5, 2e-5, f"y{i}")
    for i, (M_, K, N) in enumerate(specs):
        gy = T(f"lmy.g{i}", (M_, N))
        refs[i].backward(gy)
        outs[i].root.grad_t = gy.to(DEV).contiguous()
        outs[i].root.whole_written = True
    G.backward()
    torch.cuda.synchronize()
    for i, (xr, Wr, br, rr) in enumerate(leaves):
        close(G.grad_of(items[i][0]), xr.grad, 2e-5, 1e-4, f"dx{i}")
        close(items[i][1].grad, Wr.grad, 5e-5, 1e-4, f"dW{i}")
        close(items[i][2].grad, br.grad, 5e-5, 1e-4, f"db{i}")
        close(G.grad_of(items[i][4]), rr.grad, 1e-6, 1e-6, f"dres{i}")


def test_attention_streams_run_to_run_bit_identical():
    """scripts/attn_stress.py (short form): the three hand-scheduled streams involve no atomics, so repeated runs on the same operands must
    agree bit for bit whatever the timing (a side stream perturbs it); 30,000 runs of the long form: profiles/r4_attn_race_hunt.txt."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "attn_stress.py"), "--iters", "120", "--reseed", "30"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "RACE HUNT clean" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dropout_mask_properties(dtype):
    """engine.Graph.dropout (tc_dropout: the Dropout(0.1) of the "sp" bridge's MLP_FFN, MSTr.py:70-77).  The mask bits cannot be pinned to torch's
    generator; what is checked: kept elements are x / (1 - p) exactly, the dropped fraction is p within sampling error, the backward applies the
    SAME mask to the gradient, another seed or salt gives another mask, and outside training mode the op is the identity."""
    from transception_amd.engine import Graph, Var
    n, p = (4096, 256), 0.1
    x = (torch.rand(n, generator=torch.Generator().manual_seed(1)) + 0.5).to(dtype).to(DEV)
    gy = (torch.rand(n, generator=torch.Generator().manual_seed(2)) + 0.5).to(dtype).to(DEV)
    seed = torch.tensor([5], dtype=torch.int64, device=DEV)

    def run(seed_t, salt, training=True):
        G = Graph(dtype, torch.device(DEV), training=training, record=True)
        xv = Var(x.clone())
        out = G.dropout(xv, p, seed_t, salt)
        if out is xv:
            return None, None
        run_bwd(G, out, gy)
        return out.data.float().cpu(), G.grad_of(xv).float().cpu()
    y, dx = run(seed, 0)
    keep = y != 0
    frac = 1.0 - keep.float().mean().item()
    assert abs(frac - p) < 4 * (p * (1 - p) / keep.numel()) ** 0.5 + 1e-3, frac
    scale = torch.tensor(1.0 / (1.0 - p), dtype=torch.float32)
    want = (x.float().cpu() * scale).to(dtype).float()
    assert torch.equal(y[keep], want[keep])
    assert torch.equal(dx != 0, keep) and torch.equal(dx[keep], (gy.float().cpu() * scale).to(dtype).float()[keep])
    y2, _ = run(torch.tensor([6], dtype=torch.int64, device=DEV), 0)
    y3, _ = run(seed, 1)
    assert not torch.equal(y2 != 0, keep) and not torch.equal(y3 != 0, keep)
    assert run(seed, 0, training=False) == (None, None)


def test_sp_bridge_trains_with_dropout():
    """have_bridge = "sp" with its Dropout(0.1) live: two training forwards of the same input differ (the mask counter advances), gradients are finite,
    and the eval-mode forward is deterministic."""
    from transception_amd import MSTransception
    from transception_amd.seeded_init import schema_entries, seeded_input, seeded_labels, seeded_state_dict
    from transception_amd.train import SegLoss
    m = MSTransception(num_classes=9, have_bridge="sp")
    m.load_state_dict(seeded_state_dict(schema_entries(m)), strict=True)
    m.to(DEV).train()
    x, lab = torch.from_numpy(seeded_input(1)).to(DEV), torch.from_numpy(seeded_labels(1)).to(DEV)
    a = m(x)
    SegLoss(9)(a, lab)[0].backward()
    assert all(torch.isfinite(q.grad).all() for q in m.parameters() if q.grad is not None)
    b = m(x).detach()
    assert not torch.equal(a.detach(), b)
    m.eval()
    with torch.no_grad():
        assert torch.equal(m(x), m(x))
