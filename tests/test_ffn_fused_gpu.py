"""The fused MixFFN_skip path (engine.Graph.mixffn: LayerNorm + GELU inside the fc2 GEMMs' operand loaders / epilogue, LayerNorm
backward + depthwise gradients in tc_ffn_mid_bwd) against (a) a plain PyTorch fp32 reference of the same arithmetic
(MSTr.py:889-902 with DWConv :21-31) and (b) the unfused engine composition it replaces -- forward, input gradient and all eight
parameter gradients; single sites, stacked weight groups, and the four-site form a bridge layer uses."""
import pytest
import torch
import torch.nn.functional as F

from golden_util import gelu_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _params(C, groups, gen, pre=False):
    """One flat fp32 arena holding `groups` parameter sets at a constant stride (what the three MB encoders look like)."""
    C4 = 4 * C
    shapes = [("W1", (C4, C)), ("b1", (C4,)), ("wd", (C4, 1, 3, 3)), ("bd", (C4,)), ("lg", (C4,)), ("lb", (C4,)), ("W2", (C, C4)), ("b2", (C,))]
    if pre:
        shapes += [("pg", (C,)), ("pb", (C,))]              # the block's norm2 ahead of the site
    offs, tot = {}, 0
    for k, shp in shapes:
        offs[k] = tot
        tot += (int(torch.tensor(shp).prod()) + 7) // 8 * 8
    flat = torch.zeros(groups * tot)
    for g in range(groups):
        for k, shp in shapes:
            n = int(torch.tensor(shp).prod())
            fan = shp[1] if k in ("W1", "W2") else (9 if k == "wd" else 1)
            v = torch.randn(n, generator=gen) * (fan ** -0.5 if k in ("W1", "W2", "wd") else 0.1)
            if k in ("lg", "pg"):
                v = 1.0 + 0.2 * torch.randn(n, generator=gen)
            flat[g * tot + offs[k]: g * tot + offs[k] + n] = v
    return flat, offs, dict(shapes), tot


def _ref(x, flat, offs, shapes, tot, groups, B, H, W, res, pre_eps=None):
    """PyTorch fp32 reference per group; x [groups*B*H*W, C] token-major."""
    C = x.shape[1]
    outs = []
    M = B * H * W
    for g in range(groups):
        P = {k: flat[g * tot + offs[k]: g * tot + offs[k] + int(torch.tensor(shapes[k]).prod())].view(shapes[k]) for k in offs}
        xg = x[g * M:(g + 1) * M]
        if pre_eps is not None:
            xg = F.layer_norm(xg, (C,), P["pg"], P["pb"], pre_eps)
        h = F.linear(xg, P["W1"], P["b1"])
        hm = h.view(B, H, W, -1).permute(0, 3, 1, 2)
        d = (F.conv2d(hm, P["wd"], P["bd"], padding=1, groups=hm.shape[1]) + hm).permute(0, 2, 3, 1).reshape(M, -1)
        a = F.gelu(F.layer_norm(d, (d.shape[1],), P["lg"], P["lb"], 1e-5))
        outs.append(F.linear(a, P["W2"], P["b2"]) + res[g * M:(g + 1) * M])
    return torch.cat(outs, 0)


def _engine_run(dtype, fused, x, flat, offs, shapes, tot, groups, B, H, W, res, gout, pre_eps=None):
    import transception_amd.model as MM
    from transception_amd.engine import Graph, P, Var
    dev = torch.device(DEV)
    pf = flat.to(dev)
    pl = pf.to(dtype)
    gf = torch.zeros_like(pf)
    G = Graph(dtype, dev, training=True, record=True)
    from transception_amd import _lib
    calls0 = _lib._Lib.calls

    def mk(k):
        n = int(torch.tensor(shapes[k]).prod())
        shp = shapes[k] if k != "wd" else shapes[k]
        return P(pl[offs[k]:offs[k] + n].view(shp), gf[offs[k]:offs[k] + n].view(shp), tot if groups > 1 else 0)
    xv = Var(x.to(dev).to(dtype).contiguous())
    rv = xv if res is None else Var(res.to(dev).to(dtype).contiguous())       # res None: the block form, residual = the site's raw input
    W1, b1, wd, bd, lg, lb, W2, b2 = (mk(k) for k in ("W1", "b1", "wd", "bd", "lg", "lb", "W2", "b2"))
    pre = (mk("pg"), mk("pb"), pre_eps) if pre_eps is not None else None
    ctx = G.grouped(groups, tot) if groups > 1 else None
    if ctx:
        ctx.__enter__()
    if fused:
        out = G.mixffn([dict(x=xv, fc1=(W1, b1), dw=(wd, bd), ln=(lg, lb), fc2=(W2, b2), geo=(B, H, W), residual=rv, pre_ln=pre)])[0]
    else:
        h = G.linear(xv if pre is None else G.layernorm(xv, *pre), W1, b1)
        d = G.dwconv(h, wd, bd, B, H, W, 3, 1, True)
        a = G.layernorm(d, lg, lb, 1e-5, MM.ACT_GELU)
        out = G.linear(a, W2, b2, residual=rv)
    r = out.root
    r.grad_t = gout.to(dev).to(dtype).contiguous()
    r.whole_written = True
    G.backward()
    if ctx:
        ctx.__exit__(None, None, None)
    torch.cuda.synchronize()
    return out.data.float().cpu(), G.grad_of(xv).float().cpu(), G.grad_of(rv).float().cpu(), gf.cpu(), _lib._Lib.calls - calls0


PRE_CASES = [(64, 2, 28, 28, 1, 1e-5), (64, 2, 20, 24, 3, 1e-6), (64, 1, 56, 56, 1, 1e-5), (64, 2, 7, 9, 1, 1e-6), (128, 3, 14, 14, 1, 1e-6)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", PRE_CASES)
def test_mixffn_with_the_blocks_norm2_inside_the_tiled_kernels(case, dtype, monkeypatch):
    """out = MixFFN(LayerNorm(x)) + x (MSTr.py:168 / :945): the LayerNorm applied as the tiled kernels load x (C = 64, 16-bit types; a
    launch of its own otherwise), its backward where dx leaves -- against torch fp32 and against the engine with the fusion switched off."""
    C, B, H, W, groups, eps = case
    gen = torch.Generator().manual_seed(300 + C + H)
    flat, offs, shapes, tot = _params(C, groups, gen, pre=True)
    rows = groups * B * H * W
    x = torch.randn(rows, C, generator=gen) * 1.3 + 0.2
    gout = torch.randn(rows, C, generator=gen)
    xr, fr = x.clone().requires_grad_(True), flat.clone().requires_grad_(True)
    yr = _ref(xr, fr, offs, shapes, tot, groups, B, H, W, xr, eps)
    yr.backward(gout)
    import transception_amd.engine as E
    monkeypatch.setattr(E, "_FFN_PRE_LN", True)             # (off by default: a wash in time; kept as a switch and kept correct)
    y, gx, _, gp, nl = _engine_run(dtype, True, x, flat, offs, shapes, tot, groups, B, H, W, None, gout, eps)
    monkeypatch.setattr(E, "_FFN_PRE_LN", False)
    yu, gxu, _, gpu, nlu = _engine_run(dtype, True, x, flat, offs, shapes, tot, groups, B, H, W, None, gout, eps)
    if C == 64 and dtype != torch.float32:
        assert nl < nlu, (nl, nlu)                          # fewer C-ABI calls: the LayerNorm launches are gone
    else:
        assert nl == nlu
    tol = {torch.float32: 3e-5, torch.bfloat16: 3e-2, torch.float16: 6e-3}[dtype]

    def close(a, b, what):
        err = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-6)
        assert err < tol, f"{what}: rel-to-max error {err:.3e} (case {case}, {dtype})"
    close(y, yr.detach(), "out")
    close(gx, xr.grad, "dx")
    close(y, yu, "out vs separate LayerNorm")
    close(gx, gxu, "dx vs separate LayerNorm")
    for g in range(groups):
        for k in offs:
            n = int(torch.tensor(shapes[k]).prod())
            sl = slice(g * tot + offs[k], g * tot + offs[k] + n)
            close(gp[sl], fr.grad[sl], f"d{k}[group {g}]")
            close(gp[sl], gpu[sl], f"d{k}[group {g}] vs separate LayerNorm")


CASES = [  # C, B, H, W, groups
    (64, 2, 28, 28, 1),       # stage-2 width, tiles with overhang (28 = 16 + 12)
    (128, 3, 14, 14, 1),      # stage-3 width
    (512, 2, 7, 7, 1),        # bridge scale 4: 2048 hidden channels = 32 statistics chunks
    (320, 2, 7, 7, 1),        # 1280 hidden channels
    (64, 2, 20, 24, 3),       # three stacked weight groups, non-square map
    (64, 1, 56, 56, 1),       # stage-1 map: many tiles per workgroup walker
    (128, 2, 28, 28, 1),      # bridge scale 2 / decoder 1 width
    (128, 2, 12, 20, 3),      # C = 128 with stacked weight groups, map not a multiple of any tile
]


def _ref_rounding_model(x, flat, offs, shapes, tot, groups, B, H, W, res, dtype, pre_eps=None, stats_of_unrounded=True):
    """fp64 model of the tiled kernels' own arithmetic on storage-grid operands: h, d and a live in LDS / HBM in the storage type (rounded
    there), the LayerNorm statistics are those of the rounded d, GELU through the polynomial CDF of the 16-bit kernels (golden_util.gelu_model),
    the result is rounded once."""
    rq = lambda t: t.to(dtype).double()
    x, flat, res = rq(x), rq(flat), rq(res)
    C, M, outs = x.shape[1], B * H * W, []
    for g in range(groups):
        P = {k: flat[g * tot + offs[k]: g * tot + offs[k] + int(torch.tensor(shapes[k]).prod())].view(shapes[k]) for k in offs}
        xg = x[g * M:(g + 1) * M]
        if pre_eps is not None:
            xg = rq(F.layer_norm(xg, (C,), P["pg"], P["pb"], pre_eps))
        h = rq(F.linear(xg, P["W1"], P["b1"]))
        hm = h.view(B, H, W, -1).permute(0, 3, 1, 2)
        du = (F.conv2d(hm, P["wd"], P["bd"], padding=1, groups=hm.shape[1]) + hm).permute(0, 2, 3, 1).reshape(M, -1)
        d = rq(du)
        if stats_of_unrounded:                     # the tiled forward takes the row statistics from the fp32 values before they are rounded into LDS
            mu, var = du.mean(1, keepdim=True), du.var(1, unbiased=False, keepdim=True)
            a = rq(gelu_model((d - mu) * torch.rsqrt(var + 1e-5) * P["lg"] + P["lb"], dtype))
        else:
            a = rq(gelu_model(F.layer_norm(d, (d.shape[1],), P["lg"], P["lb"], 1e-5), dtype))
        outs.append(rq(F.linear(a, P["W2"], P["b2"]) + res[g * M:(g + 1) * M]))
    return torch.cat(outs, 0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [(64, 2, 28, 28, 1), (64, 1, 56, 56, 1), (64, 2, 7, 9, 1), (128, 3, 14, 14, 1), (128, 2, 28, 28, 3)])
def test_fused_mixffn_forward_within_one_rounding_of_its_fp64_model(case, dtype):
    """ffn_fused_fwd_kernel on storage-grid operands against the fp64 model of its own arithmetic, <= 3e-3 (bf16) /
    1e-3 (fp16) of the result's largest value -- the 3e-2 of the fp32 comparison below would let a wrong tap or bias through."""
    C, B, H, W, groups = case
    gen = torch.Generator().manual_seed(300 + C + H)
    flat, offs, shapes, tot = _params(C, groups, gen)
    rows = groups * B * H * W
    x, res, gout = (torch.randn(rows, C, generator=gen) for _ in range(3))
    y, *_ = _engine_run(dtype, True, x, flat, offs, shapes, tot, groups, B, H, W, res, gout)
    ym = _ref_rounding_model(x, flat, offs, shapes, tot, groups, B, H, W, res, dtype)
    err = float((y.double() - ym).abs().max() / ym.abs().max())
    ym2 = _ref_rounding_model(x, flat, offs, shapes, tot, groups, B, H, W, res, dtype, stats_of_unrounded=False)
    err2 = float((y.double() - ym2).abs().max() / ym2.abs().max())
    print(f"ffn_fused_fwd {case} {dtype}: out vs the fp64 rounding model {err:.2e} (statistics of the rounded d: {err2:.2e})")
    # fp16: one spacing of the largest results (2^-10 of their binade): the kernel's fp32 value and the model's fp64 value of an element can sit on
    # either side of a rounding boundary (about one element in 4000 does), which costs a whole spacing, not half of one
    assert err < {torch.bfloat16: 3e-3, torch.float16: 1e-3}[dtype], err


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CASES)
def test_fused_mixffn_matches_torch_and_unfused(case, dtype):
    C, B, H, W, groups = case
    gen = torch.Generator().manual_seed(100 + C + H)
    flat, offs, shapes, tot = _params(C, groups, gen)
    rows = groups * B * H * W
    x = torch.randn(rows, C, generator=gen)
    res = torch.randn(rows, C, generator=gen)
    gout = torch.randn(rows, C, generator=gen)
    # PyTorch reference (fp32, CPU) with autograd
    xr, fr, rr = x.clone().requires_grad_(True), flat.clone().requires_grad_(True), res.clone().requires_grad_(True)
    yr = _ref(xr, fr, offs, shapes, tot, groups, B, H, W, rr)
    yr.backward(gout)
    y, gx, gr, gp, _ = _engine_run(dtype, True, x, flat, offs, shapes, tot, groups, B, H, W, res, gout)
    tol = {torch.float32: 3e-5, torch.bfloat16: 3e-2, torch.float16: 6e-3}[dtype]

    def close(a, b, what):
        scale = float(b.abs().max()) + 1e-6
        err = float((a - b).abs().max()) / scale
        assert err < tol, f"{what}: rel-to-max error {err:.3e} (case {case}, {dtype})"
    close(y, yr.detach(), "out")
    close(gx, xr.grad, "dx")
    close(gr, rr.grad, "dresidual")
    for g in range(groups):
        for k in offs:
            n = int(torch.tensor(shapes[k]).prod())
            sl = slice(g * tot + offs[k], g * tot + offs[k] + n)
            close(gp[sl], fr.grad[sl], f"d{k}[group {g}]")
    # and the unfused engine composition (same storage type): the two must agree at least as closely
    yu, gxu, gru, gpu, _ = _engine_run(dtype, False, x, flat, offs, shapes, tot, groups, B, H, W, res, gout)
    close(y, yu, "out vs unfused")
    close(gx, gxu, "dx vs unfused")
    close(gp, gpu, "parameter gradients vs unfused")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_fused_mixffn_four_sites_one_launch_set(dtype):
    """The bridge form: four independent sites of different widths / map sizes in one fused call (3 + 3 launches)."""
    from transception_amd.engine import Graph, P, Var
    dev = torch.device(DEV)
    gen = torch.Generator().manual_seed(7)
    B = 2
    geo = [(64, 56), (128, 28), (320, 14), (512, 7)]
    G = Graph(dtype, dev, training=True, record=True)
    sites, refs, keep = [], [], []
    for C, S in geo:
        flat, offs, shapes, tot = _params(C, 1, gen)
        rows = B * S * S
        x, res, gout = (torch.randn(rows, C, generator=gen) for _ in range(3))
        pf = flat.to(dev); pl = pf.to(dtype); gf = torch.zeros_like(pf)
        mk = lambda k: P(pl[offs[k]:offs[k] + int(torch.tensor(shapes[k]).prod())].view(shapes[k]),
                         gf[offs[k]:offs[k] + int(torch.tensor(shapes[k]).prod())].view(shapes[k]), 0)
        xv, rv = Var(x.to(dev).to(dtype)), Var(res.to(dev).to(dtype))
        sites.append(dict(x=xv, fc1=(mk("W1"), mk("b1")), dw=(mk("wd"), mk("bd")), ln=(mk("lg"), mk("lb")), fc2=(mk("W2"), mk("b2")),
                          geo=(B, S, S), residual=rv))
        xr, fr = x.clone().requires_grad_(True), flat.clone().requires_grad_(True)
        yr = _ref(xr, fr, offs, shapes, tot, 1, B, S, S, res)
        yr.backward(gout)
        refs.append((yr.detach(), xr.grad, fr.grad))
        keep.append((xv, gf, gout))
    n0 = G.n_launch
    outs = G.mixffn(sites)
    # fp32: fc1 x4 and fc2 x4, one merged grid each (+ one conv launch); 16-bit: the C = 64 / 128 sites are one tiled kernel each
    assert G.n_launch - n0 == (2 if dtype == torch.float32 else 4)
    for o, (_, _, gout) in zip(outs, keep):
        o.root.grad_t = gout.to(dev).to(dtype).contiguous()
        o.root.whole_written = True
    G.backward()
    torch.cuda.synchronize()
    tol = {torch.float32: 3e-5, torch.bfloat16: 3e-2, torch.float16: 6e-3}[dtype]
    for o, (xv, gf, _), (yr, gxr, gpr) in zip(outs, keep, refs):
        for a, b in ((o.data.float().cpu(), yr), (G.grad_of(xv).float().cpu(), gxr), (gf.cpu(), gpr)):
            assert float((a - b).abs().max()) / (float(b.abs().max()) + 1e-6) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_mixffn_recompute_and_wide_loader_variants(dtype, monkeypatch):
    """The library paths the default configuration does not take: LayerNorm + GELU in the fc2 loader of a multi-tile product
    (TC_FFN_LN_A with N > 64) and the weight gradient that recomputes GELU(LN(d)) in its B loader (TC_FFN_LN_B) instead of reading
    the stored activation."""
    import transception_amd.engine as E
    monkeypatch.setattr(E, "_FFN_STORE_ACT", False)
    monkeypatch.setattr(E, "_FFN_LN_GEMM_MAXC", 4096)
    for case in ((128, 3, 14, 14, 1), (64, 2, 20, 24, 3), (320, 2, 7, 7, 1)):
        test_fused_mixffn_matches_torch_and_unfused(case, dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tile", [(14, 8), (7, 8), (3, 28), (5, 7), (4, 6), (1, 2), (7, 7), (4, 14), (2, 30)])
def test_tiled_forward_kernel_tile_shapes(tile, dtype, monkeypatch):
    """csrc/mixffn.hip with forced pixel tiles: even / odd widths, tiles hanging over the map edge, one-row tiles, tiles wider than the
    map -- the result must not depend on the tiling."""
    import transception_amd.engine as E
    th, tw = tile
    for C, B, H, W, groups in ((64, 2, 28, 28, 1), (64, 1, 9, 30, 2), (128, 2, 14, 14, 1), (128, 1, 7, 7, 3)):
        if (th + 2) * (tw + 2) > (160 if C == 64 else 96) or th * tw > (128 if C == 64 else 64):
            continue
        monkeypatch.setattr(E, "_FFN_TILE", (th, tw))
        test_fused_mixffn_matches_torch_and_unfused((C, B, H, W, groups), dtype)
