"""Fused pieces of the MHCABlock (MSTr.py:826-946) against plain PyTorch fp32 restatements and against the op-by-op launches they
replace.  16-bit storage only (the fp32 parity path keeps the op-by-op form)."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transception_amd.seeded_init import seeded_tensor  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
WINDOWS = [(3, 2), (5, 3), (7, 3)]


def T(tag, shape, scale=1.0):
    return torch.from_numpy(seeded_tensor("mhca/" + tag, shape, scale))


def rel(got, want):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    return (got - want).abs().max().item() / (want.abs().max().item() + 1e-12)


# the fused 16-bit kernels against an fp64 model of their OWN arithmetic -- operands on the storage grid, every value the
# kernel stores rounded where the kernel rounds it -- within about one rounding of the result's largest value: half an ulp of the top
# binade is 2^-9 = 2.0e-3 (bf16) / 2^-12 = 2.4e-4 (fp16), and an intermediate that rounds the other way than the model's (fp32 against
# fp64 sums near a tie) moves a result by one more; measured on MI355X: <= 1.8e-3 / 5.1e-4.  A wrong halo row, tap or bias moves an output by >= 1e-2.
TIGHT = {torch.bfloat16: 3e-3, torch.float16: 8e-4}


def rq(t, dtype):
    """Round to the storage type, continue in fp64."""
    return t.to(dtype).double()


def _arena(C, Gn, dtype, tag):
    """The parameters of Gn weight groups laid out at a constant stride, as in the model's flat arenas: Wqkv | bqkv | w3 | b3 | w5 | b5 | w7 | b7."""
    from transception_amd.engine import P
    Ch = C // 8
    shapes = [(3 * C, C), (3 * C,)]
    for k, nh in WINDOWS:
        shapes += [(nh * Ch, k * k), (nh * Ch,)]
    sizes = [(int(torch.tensor(s).prod()) + 7) // 8 * 8 for s in shapes]
    per = sum(sizes)
    flat = T(tag, (Gn * per,), 0.15)
    lp = flat.to(dtype).to(DEV)
    master = lp.float().cpu()                                     # the reference sees the rounded parameters
    gflat = torch.zeros(Gn * per, dtype=torch.float32, device=DEV)
    Ps, off = [], 0
    for s, n in zip(shapes, sizes):
        ne = int(torch.tensor(s).prod())
        Ps.append(P(lp[off:off + ne].view(s), gflat[off:off + ne].view(s), per if Gn > 1 else 0))
        off += n
    return Ps, master, gflat, per, shapes, sizes


def _reference(x, master, per, shapes, sizes, Gn, B, side, C, rnd=None):
    """rnd: the rounding model (fp64 in, storage-grid fp64 out) applied where tc_mhca_att_fwd stores -- q | k | v, crpe(v), the result;
    None: the plain fp32 statement with autograd."""
    Ch, N = C // 8, side * side
    if rnd is not None:
        x, master = x.double(), master.double()
    pr = master.clone().requires_grad_(rnd is None)
    xr = x.clone().requires_grad_(rnd is None)
    rnd = rnd or (lambda t: t)
    outs = []
    for g in range(Gn):
        ps, off = [], g * per
        for s, n in zip(shapes, sizes):
            ne = int(torch.tensor(s).prod())
            ps.append(pr[off:off + ne].view(s)); off += n
        xg = xr[g * B * N:(g + 1) * B * N]
        qkv = rnd(xg @ ps[0].t() + ps[1])
        q, k, v = (qkv[:, i * C:(i + 1) * C] for i in range(3))
        vim = v.reshape(B, side, side, C).permute(0, 3, 1, 2)
        cs, c0 = [], 0
        for i, (ks, nh) in enumerate(WINDOWS):
            wd = nh * Ch
            cs.append(F.conv2d(vim[:, c0:c0 + wd], ps[2 + 2 * i].view(wd, 1, ks, ks), ps[3 + 2 * i], padding=ks // 2, groups=wd)); c0 += wd
        convv = rnd(torch.cat(cs, 1).permute(0, 2, 3, 1).reshape(B * N, C))
        qh, kh, vh = (t.reshape(B, N, 8, Ch).permute(0, 2, 1, 3) for t in (q, k, v))
        ctx = torch.softmax(kh, dim=2).transpose(-1, -2) @ vh
        fa = (qh @ ctx).permute(0, 2, 1, 3).reshape(B * N, C)
        outs.append(rnd(Ch ** -0.5 * fa + q * convv))
    return torch.cat(outs, 0), xr, pr


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("C,side,B,Gn", [(64, 28, 2, 3), (128, 14, 2, 3), (320, 7, 3, 3), (128, 14, 1, 1), (64, 12, 2, 1), (320, 5, 2, 1)])
def test_mhca_attention_one_launch(dtype, C, side, B, Gn):
    """engine.Graph.mhca_attention (tc_mhca_att_fwd: qkv projection + ConvRelPosEnc + factorised attention per (image, head)) against the
    fp32 torch statement of MSTr.py:852-886, forward and -- through the recorded closures of the ops it replaces -- backward; and against
    the three launches it replaces on the same operands."""
    import transception_amd.engine as E
    from transception_amd.engine import Graph, Var
    N, Ch = side * side, C // 8
    rows = Gn * B * N
    x16 = T(f"x{C}.{side}", (rows, C)).to(dtype)
    gy16 = T(f"g{C}.{side}", (rows, C)).to(dtype)
    Ps, master, gflat, per, shapes, sizes = _arena(C, Gn, dtype, f"p{C}.{Gn}")
    ref, xr, pr = _reference(x16.float(), master, per, shapes, sizes, Gn, B, side, C)
    ref.backward(gy16.float())

    def run(fused, fused_bwd=True):
        E._MHCA_ATT_FUSED = fused
        E._MHCA_ATT_BWD_FUSED = fused_bwd
        gflat.zero_()
        G = Graph(dtype, torch.device(DEV), training=True, record=True)
        xv = Var(x16.to(DEV).contiguous())
        assert G.mhca_att_supported(xv, N) == fused
        ctxm = G.grouped(Gn, per) if Gn > 1 else None
        if ctxm is not None:
            ctxm.__enter__()
        if fused:
            o = G.mhca_attention(xv, Ps[0], Ps[1], [Ps[2], Ps[4], Ps[6]], [Ps[3], Ps[5], Ps[7]], B, side, 8, Ch ** -0.5, WINDOWS)
        else:
            qkv = G.linear(xv, Ps[0], Ps[1], out=G.new(rows, 3 * C, covered=True))
            q, k, v = qkv.colslice(0, C), qkv.colslice(C, 2 * C), qkv.colslice(2 * C, 3 * C)
            convv = G.new(rows, C)
            c0, xs, outs = 0, [], []
            for ks, nh in WINDOWS:
                xs.append(v.colslice(c0, c0 + nh * Ch)); outs.append(convv.colslice(c0, c0 + nh * Ch)); c0 += nh * Ch
            G.dwconv_multi(xs, [Ps[2], Ps[4], Ps[6]], [Ps[3], Ps[5], Ps[7]], (B, side, side), [3, 5, 7], outs)
            o = G.factor_att_core(q, k, v, convv, Gn * B, N, 8, Ch ** -0.5)
        o.root.grad_t = gy16.to(DEV).contiguous()
        o.root.whole_written = True
        G.backward()
        nl = G.n_launch
        if ctxm is not None:
            ctxm.__exit__(None, None, None)
        torch.cuda.synchronize()
        return o.data.float().cpu(), G.grad_of(xv).float().cpu(), gflat.cpu().clone(), nl

    try:
        of, dxf, gpf, nlf = run(True)
        o2, dx2, gp2, _ = run(True, fused_bwd=False)
        ou, dxu, gpu_, nlu = run(False)
    finally:
        E._MHCA_ATT_FUSED = E._MHCA_ATT_BWD_FUSED = True
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert rel(of, ref) < tol, ("o vs torch", rel(of, ref))
    with torch.no_grad():
        ref_m = _reference(x16.float(), master, per, shapes, sizes, Gn, B, side, C, rnd=lambda t: rq(t, dtype))[0]
    print(f"mhca_att_fwd C={C} {dtype}: o vs the fp64 rounding model {rel(of, ref_m):.2e} (vs plain fp32 {rel(of, ref):.2e})")
    assert rel(of, ref_m) < TIGHT[dtype], ("o vs the fp64 rounding model", rel(of, ref_m))
    assert rel(dxf, xr.grad) < 2 * tol, ("dx vs torch", rel(dxf, xr.grad))
    assert rel(gpf, pr.grad) < 2 * tol, ("parameter gradients vs torch", rel(gpf, pr.grad))
    # against the op-by-op launches: the same roundings at the same places, only the summation order of the projection differs
    assert rel(of, ou) < tol / 2, ("o fused vs unfused", rel(of, ou))
    assert rel(dxf, dxu) < tol, ("dx fused vs unfused", rel(dxf, dxu))
    assert rel(gpf, gpu_) < tol, ("parameter gradients fused vs unfused", rel(gpf, gpu_))
    # the one-launch backward against the two launches it replaces (same forward)
    assert rel(dxf, dx2) < tol / 2, ("dx, fused backward vs factor_att_bwd + dwconv_multi", rel(dxf, dx2))
    assert rel(gpf, gp2) < tol / 2, ("parameter gradients, fused backward vs factor_att_bwd + dwconv_multi", rel(gpf, gp2))
    assert nlf == 3 and nlu >= 1                      # forward, fused backward, the projection's gradient pair


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("C,side,B,Gn", [(64, 28, 2, 3), (128, 14, 2, 3), (320, 7, 3, 3), (128, 9, 1, 1)])
def test_dw_ln_one_launch(dtype, C, side, B, Gn):
    """engine.Graph.dw_ln (tc_dw_ln_fwd: ConvPosEnc dw3x3 + skip and norm1 of an MHCABlock in one launch) against torch fp32 and against the
    two launches it replaces (forward identical up to the summation order; backward is theirs)."""
    import transception_amd.engine as E
    from transception_amd.engine import Graph, P, Var
    rows = Gn * B * side * side
    x16, g1, g2 = T(f"dl.x{C}", (rows, C)).to(dtype), T(f"dl.g1{C}", (rows, C)).to(dtype), T(f"dl.g2{C}", (rows, C)).to(dtype)
    per = (9 * C + C + C + C + 7) // 8 * 8
    flat = T(f"dl.p{C}", (Gn * per,), 0.3)
    flat.view(Gn, per)[:, 10 * C:11 * C] += 1.0                                   # gamma around 1
    lp = flat.to(dtype).to(DEV)
    master = lp.float().cpu().clone().requires_grad_()
    gflat = torch.zeros(Gn * per, dtype=torch.float32, device=DEV)
    mk = lambda a, n, shp: P(lp[a:a + n].view(shp), gflat[a:a + n].view(shp), per if Gn > 1 else 0)
    w, b, ga, be = mk(0, 9 * C, (C, 9)), mk(9 * C, C, (C,)), mk(10 * C, C, (C,)), mk(11 * C, C, (C,))
    xr = x16.float().requires_grad_()
    outs_t, outs_n = [], []
    for g in range(Gn):
        m = master[g * per:(g + 1) * per]
        xi = xr[g * B * side * side:(g + 1) * B * side * side].reshape(B, side, side, C).permute(0, 3, 1, 2)
        t1 = (F.conv2d(xi, m[:9 * C].view(C, 1, 3, 3), m[9 * C:10 * C], padding=1, groups=C) + xi).permute(0, 2, 3, 1).reshape(-1, C)
        outs_t.append(t1); outs_n.append(F.layer_norm(t1, (C,), m[10 * C:11 * C], m[11 * C:12 * C], 1e-6))
    rt, rn = torch.cat(outs_t), torch.cat(outs_n)
    (rt * g1.float()).sum().backward(retain_graph=True)
    (rn * g2.float()).sum().backward()

    def run(fused):
        E._DW_LN_FUSED = fused
        gflat.zero_()
        G = Graph(dtype, torch.device(DEV), training=True, record=True)
        xv = Var(x16.to(DEV).contiguous())
        with (G.grouped(Gn, per) if Gn > 1 else G.grouped(1, 0)):
            assert G.dw_ln_supported(xv) == fused
            if fused:
                t1, xn = G.dw_ln(xv, w, b, ga, be, B, side, side, 1e-6)
            else:
                t1 = G.dwconv(xv, w, b, B, side, side, 3, 1, True)
                xn = G.layernorm(t1, ga, be, 1e-6)
            xn.root.grad_t = g2.to(DEV).contiguous(); xn.root.whole_written = True
            t1.root.grad_t = g1.to(DEV).contiguous(); t1.root.whole_written = True       # (the residual branch's gradient of t1: LayerNorm's backward adds to it)
            G.backward()
        torch.cuda.synchronize()
        return t1.data.float().cpu(), xn.data.float().cpu(), G.grad_of(xv).float().cpu(), gflat.cpu().clone()
    try:
        tf, nf, dxf, gpf = run(True)
        tu, nu, dxu, gpu_ = run(False)
    finally:
        E._DW_LN_FUSED = True
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert rel(tf, rt) < tol and rel(nf, rn) < tol, (rel(tf, rt), rel(nf, rn))
    with torch.no_grad():                                                          # fp64 rounding model: t1 rounded, norm1 of the ROUNDED t1, rounded
        mt, mn = [], []
        md = master.detach().double()
        for g in range(Gn):
            m = md[g * per:(g + 1) * per]
            xi = x16.double()[g * B * side * side:(g + 1) * B * side * side].reshape(B, side, side, C).permute(0, 3, 1, 2)
            t1 = rq((F.conv2d(xi, m[:9 * C].view(C, 1, 3, 3), m[9 * C:10 * C], padding=1, groups=C) + xi).permute(0, 2, 3, 1).reshape(-1, C), dtype)
            mt.append(t1); mn.append(rq(F.layer_norm(t1, (C,), m[10 * C:11 * C], m[11 * C:12 * C], 1e-6), dtype))
        mt, mn = torch.cat(mt), torch.cat(mn)
    print(f"dw_ln_fwd C={C} {dtype}: t1 / norm1 vs the fp64 rounding model {rel(tf, mt):.2e} / {rel(nf, mn):.2e}")
    assert rel(tf, mt) < TIGHT[dtype] and rel(nf, mn) < TIGHT[dtype], (rel(tf, mt), rel(nf, mn))
    assert rel(tf, tu) < tol / 4 and rel(nf, nu) < tol / 2, (rel(tf, tu), rel(nf, nu))
    assert rel(dxf, xr.grad) < 2 * tol and rel(gpf, master.grad) < 2 * tol, (rel(dxf, xr.grad), rel(gpf, master.grad))
    assert rel(dxf, dxu) < tol and rel(gpf, gpu_) < tol


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("C,rows_g,Gn,res", [(64, 2 * 784, 3, True), (128, 2 * 196, 3, True), (320, 3 * 49, 3, True), (64, 1029, 1, True),
                                             (128, 81, 1, False), (320, 200, 1, True)])
def test_linear_ln_one_launch(dtype, C, rows_g, Gn, res):
    """engine.Graph.linear_ln (tc_linear_ln_fwd: square Linear + bias + skip and the LayerNorm after it in one launch -- proj + norm2 of an
    MHCABlock / bridge layer, reprojection + norm2 of an EfficientTransformerBlock) against torch fp32 and against the two launches it
    replaces (forward: same roundings at the same places; backward is theirs).  Row counts include a partial last 64-row tile per group."""
    import transception_amd.engine as E
    from transception_amd.engine import Graph, P, Var
    rows = Gn * rows_g
    x16, r16 = T(f"ll.x{C}", (rows, C)).to(dtype), T(f"ll.r{C}", (rows, C)).to(dtype)
    g1, g2 = T(f"ll.g1{C}", (rows, C)).to(dtype), T(f"ll.g2{C}", (rows, C)).to(dtype)
    per = (C * C + 3 * C + 7) // 8 * 8
    flat = T(f"ll.p{C}", (Gn * per,), C ** -0.5)
    flat.view(Gn, per)[:, C * C + C:C * C + 2 * C] += 1.0                          # gamma around 1
    lp = flat.to(dtype).to(DEV)
    master = lp.float().cpu().clone().requires_grad_()
    gflat = torch.zeros(Gn * per, dtype=torch.float32, device=DEV)
    mk = lambda a, n, shp: P(lp[a:a + n].view(shp), gflat[a:a + n].view(shp), per if Gn > 1 else 0)
    W, b, ga, be = mk(0, C * C, (C, C)), mk(C * C, C, (C,)), mk(C * C + C, C, (C,)), mk(C * C + 2 * C, C, (C,))
    xr, rr = x16.float().requires_grad_(), r16.float().requires_grad_()
    outs_t, outs_n = [], []
    for g in range(Gn):
        m = master[g * per:(g + 1) * per]
        t = F.linear(xr[g * rows_g:(g + 1) * rows_g], m[:C * C].view(C, C), m[C * C:C * C + C])
        if res:
            t = t + rr[g * rows_g:(g + 1) * rows_g]
        outs_t.append(t); outs_n.append(F.layer_norm(t, (C,), m[C * C + C:C * C + 2 * C], m[C * C + 2 * C:C * C + 3 * C], 1e-6))
    rt, rn = torch.cat(outs_t), torch.cat(outs_n)
    (rt * g1.float()).sum().backward(retain_graph=True)
    (rn * g2.float()).sum().backward()

    def run(fused):
        E._LIN_LN_FUSED = fused
        gflat.zero_()
        G = Graph(dtype, torch.device(DEV), training=True, record=True)
        xv, rv = Var(x16.to(DEV).contiguous()), (Var(r16.to(DEV).contiguous()) if res else None)
        with (G.grouped(Gn, per) if Gn > 1 else G.grouped(1, 0)):
            assert G.linear_ln_supported(xv, W, rv) == fused
            n0 = G.n_launch
            if fused:
                t, xn = G.linear_ln(xv, W, b, rv, ga, be, 1e-6)
                assert G.n_launch - n0 == 1
            else:
                t = G.linear(xv, W, b, residual=rv)
                xn = G.layernorm(t, ga, be, 1e-6)
            xn.root.grad_t = g2.to(DEV).contiguous(); xn.root.whole_written = True
            t.root.grad_t = g1.to(DEV).contiguous(); t.root.whole_written = True       # (the skip branch's gradient of t: LayerNorm's backward adds to it)
            G.backward()
        torch.cuda.synchronize()
        dr = G.grad_of(rv).float().cpu() if res else None
        return t.data.float().cpu(), xn.data.float().cpu(), G.grad_of(xv).float().cpu(), dr, gflat.cpu().clone()
    try:
        tf, nf, dxf, drf, gpf = run(True)
        tu, nu, dxu, dru, gpu_ = run(False)
    finally:
        E._LIN_LN_FUSED = True
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert rel(tf, rt) < tol and rel(nf, rn) < tol, (rel(tf, rt), rel(nf, rn))
    with torch.no_grad():                                                          # fp64 rounding model: t rounded before the statistics (what the backward reads)
        mt, mn = [], []
        md = master.detach().double()
        for g in range(Gn):
            m = md[g * per:(g + 1) * per]
            t = F.linear(x16.double()[g * rows_g:(g + 1) * rows_g], m[:C * C].view(C, C), m[C * C:C * C + C])
            if res:
                t = t + r16.double()[g * rows_g:(g + 1) * rows_g]
            t = rq(t, dtype)
            mt.append(t); mn.append(rq(F.layer_norm(t, (C,), m[C * C + C:C * C + 2 * C], m[C * C + 2 * C:C * C + 3 * C], 1e-6), dtype))
        mt, mn = torch.cat(mt), torch.cat(mn)
    print(f"lin_res_ln C={C} {dtype}: t / LayerNorm(t) vs the fp64 rounding model {rel(tf, mt):.2e} / {rel(nf, mn):.2e}")
    assert rel(tf, mt) < TIGHT[dtype] and rel(nf, mn) < TIGHT[dtype], (rel(tf, mt), rel(nf, mn))
    assert rel(tf, tu) < tol / 4 and rel(nf, nu) < tol / 2, (rel(tf, tu), rel(nf, nu))
    assert rel(dxf, xr.grad) < 2 * tol and rel(gpf, master.grad) < 2 * tol, (rel(dxf, xr.grad), rel(gpf, master.grad))
    assert rel(dxf, dxu) < tol and rel(gpf, gpu_) < tol
    if res:
        assert rel(drf, rr.grad) < 2 * tol and rel(drf, dru) < tol
