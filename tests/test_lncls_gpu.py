"""FinalPatchExpand_X4's rearrange + LayerNorm and last_layer as one launch each way (engine.Graph.ln_cls, csrc/lncls.hip; MSTr.py:222-225, 281)
against a plain PyTorch statement with autograd, against an fp64 model of the kernel's own arithmetic (operands on the storage grid, one
rounding of the result), and against the launches it replaces (tc_layernorm_ps_fwd + tc_gemm; tc_gemm_pair + tc_layernorm_ps_bwd)."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transception_amd.seeded_init import seeded_tensor  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TIGHT = {torch.bfloat16: 3e-3, torch.float16: 8e-4}


def T(tag, shape, scale=1.0):
    return torch.from_numpy(seeded_tensor("lncls/" + tag, shape, scale))


def rel(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    return (got - want).abs().max().item() / (want.abs().max().item() + 1e-12)


def shuffle(x, B, H, W, p, c):
    """'b h w (p1 p2 c) -> b (h p1) (w p2) c' on the token matrix [B*H*W, p*p*c]"""
    return x.view(B, H, W, p, p, c).permute(0, 1, 3, 2, 4, 5).reshape(B * H * p * W * p, c)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("B,side,ncls,pad", [(2, 8, 9, True), (2, 8, 9, False), (1, 12, 2, True), (3, 5, 2, False), (2, 14, 9, True)])
def test_ln_cls_one_launch_each_way(dtype, B, side, ncls, pad):
    import transception_amd.engine as E
    from transception_amd.engine import Graph, P, Var
    p, c = 4, 64
    rows_in, rows = B * side * side, B * side * side * p * p
    x16 = T(f"x{B}.{side}", (rows_in, p * p * c)).to(dtype)
    gl16 = T(f"gl{B}.{side}.{ncls}", (rows, ncls), 0.5).to(dtype)
    per = (2 * c + ncls * c + ncls + 7) // 8 * 8 + 8
    flat = T(f"p{ncls}", (per,), 0.3)
    flat[:c] += 1.0                                                         # gamma around 1
    lp = flat.to(dtype).to(DEV)
    master = lp.float().cpu().clone().requires_grad_()
    gflat = torch.zeros(per, dtype=torch.float32, device=DEV)
    o_w, o_b = 2 * c, 2 * c + (ncls * c + 7) // 8 * 8                       # (the classifier weight starts 16-byte aligned, its bias anywhere)
    mk = lambda a, n, shp: P(lp[a:a + n].view(shp), gflat[a:a + n].view(shp), 0)
    ga, be, Wc, bc = mk(0, c, (c,)), mk(c, c, (c,)), mk(o_w, ncls * c, (ncls, c)), mk(o_b, ncls, (ncls,))

    def torch_ref(xin, m, dt=None):
        rq = (lambda t: t) if dt is None else (lambda t: t.to(dt).double())
        xs = shuffle(xin, B, side, side, p, c)
        xn = F.layer_norm(xs, (c,), m[:c], m[c:2 * c], 1e-5)
        return rq(F.linear(xn, m[o_w:o_w + ncls * c].view(ncls, c), m[o_b:o_b + ncls]))
    xr = x16.float().requires_grad_()
    ref = torch_ref(xr, master)
    (ref * gl16.float()).sum().backward()
    with torch.no_grad():
        ref_m = torch_ref(x16.double(), master.detach().double(), dtype)    # the fused kernel keeps xn in fp32: ONE rounding, of the logits

    def run(fused):
        E._LN_CLS_FUSED = fused
        gflat.zero_()
        G = Graph(dtype, torch.device(DEV), training=True, record=True)
        xv = Var(x16.to(DEV).contiguous())
        assert G.ln_cls_supported(xv, p, ga, be, Wc, bc) == fused
        n0 = G.n_launch
        if fused:
            lg = G.ln_cls(xv, ga, be, Wc, bc, B, side, side, p, pad_rows=pad)
            assert G.n_launch - n0 == 1 and lg.ld == ((ncls + 7) // 8 * 8 if pad else ncls)
        else:
            lg = G.linear(G.layernorm_shuffled(xv, ga, be, B, side, side, p), Wc, bc)
        full = torch.full(lg.root.data.shape, float("nan"), dtype=dtype, device=DEV)     # pad columns of the gradient hold anything
        lg.apply_path(full).copy_(gl16.to(DEV))
        lg.root.grad_t = full; lg.root.whole_written = True
        G.backward()
        torch.cuda.synchronize()
        return lg.data.float().cpu(), G.grad_of(xv).float().cpu(), gflat.cpu().clone()
    try:
        lf, dxf, gpf = run(True)
        lu, dxu, gpu_ = run(False)
    finally:
        E._LN_CLS_FUSED = True
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    print(f"ln_cls B={B} side={side} ncls={ncls} pad={pad} {dtype}: logits vs the fp64 rounding model {rel(lf, ref_m):.2e}, vs op-by-op {rel(lf, lu):.2e}; "
          f"dx vs torch {rel(dxf, xr.grad):.2e} (op-by-op {rel(dxu, xr.grad):.2e}); parameter gradients vs torch {rel(gpf, master.grad):.2e} (op-by-op {rel(gpu_, master.grad):.2e})")
    assert rel(lf, ref_m) < TIGHT[dtype], rel(lf, ref_m)
    assert rel(lf, ref) < tol and rel(lf, lu) < tol
    assert rel(dxf, xr.grad) < tol and rel(dxf, dxu) < tol, (rel(dxf, xr.grad), rel(dxf, dxu))
    # parameter gradients: fp32 sums of products of storage-grid operands with fp32 xn -- tighter than the op-by-op form (which rounds xn)
    assert rel(gpf, master.grad) < 2e-3, rel(gpf, master.grad)
    assert rel(gpf, gpu_) < tol
    for a, n in ((0, c), (c, c), (o_w, ncls * c), (o_b, ncls)):            # each parameter on its own scale
        assert rel(gpf[a:a + n], master.grad[a:a + n]) < 5e-3, (a, rel(gpf[a:a + n], master.grad[a:a + n]))
