"""Fused EfficientAttention block (tc_effatt_fwd / tc_effatt_bwd) against the pinned CPU oracle's statement of
MSTr.py:80-143 + :166-167  (tx = x + attn(norm1(x)), one head), evaluated on the stored (rounded) inputs in fp32.  Tolerances: the maps are stored in bf16 / fp16
(2^-8 / 2^-11 relative) and pass through five chained 64-deep products; 2 % (bf16) and 0.5 % (fp16) of the
reference's maximum per tensor, stated per assert."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(t, P, B, N, eps=1e-5):
    """tx = x + attn(norm1(x)) through the pinned CPU oracle (oracle/transception_oracle.py: layernorm + efficient_attention, the
    restatement of MSTr.py:80-143 that tests/test_oracle_golden.py holds to the reference's own outputs) -- one source of truth."""
    from oracle.transception_oracle import TransCeptionOracle
    c = t.shape[1]
    names = {"gamma": "blk.norm1.weight", "beta": "blk.norm1.bias", "wk": "blk.attn.keys.weight", "bk": "blk.attn.keys.bias",
             "wq": "blk.attn.queries.weight", "bq": "blk.attn.queries.bias", "wv": "blk.attn.values.weight", "bv": "blk.attn.values.bias",
             "wr": "blk.attn.reprojection.weight", "br": "blk.attn.reprojection.bias"}
    orc = TransCeptionOracle({names[k]: v for k, v in P.items()}, 9, training=True)
    t3 = t.view(B, N, c)
    out = t3 + orc.efficient_attention(orc.layernorm(t3, "blk.norm1", eps), "blk.attn")
    return out.reshape(B * N, c), orc.taps["blk.attn.ctx"]


NAMES = ["gamma", "beta", "wk", "bk", "wq", "bq", "wv", "bv", "wr", "br"]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,N,pad", [(2, 3136, 0), (3, 200, 0), (1, 32, 0), (2, 257, 8), (1, 5, 0), (2, 784, 24)])
def test_effatt_fused_forward_and_backward_vs_torch_fp32(dtype, B, N, pad):
    """pad > 0: the token maps are column slices of wider buffers (row pitch C + pad), as the engine's views can be."""
    from transception_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    c = 64
    g = torch.Generator(device="cpu").manual_seed(5 + N)
    P32 = {}
    for n in NAMES:
        shape = (c, c) if n.startswith("w") else (c,)
        scale = 0.25 if n.startswith("w") else 0.3
        P32[n] = (torch.randn(shape, generator=g) * scale + (1.0 if n == "gamma" else 0.0))
    t32 = torch.randn(B * N, c, generator=g) * 1.5
    dy32 = torch.randn(B * N, c, generator=g)
    Pl = {n: P32[n].to(dev, dtype).contiguous() for n in NAMES}
    def wide(v):                                             # [rows, C] view of a [rows, C + pad] buffer
        buf = torch.full((v.shape[0], c + pad), 7.0, device=dev, dtype=dtype)
        buf[:, :c] = v.to(dev, dtype)
        return buf[:, :c]
    tl, dyl = wide(t32), wide(dy32)
    # the reference sees exactly the stored (rounded) inputs, in fp32
    Pr = {n: Pl[n].float().requires_grad_(True) for n in NAMES}
    tr = tl.float().requires_grad_(True)
    out_r, ctx_r = _ref(tr, Pr, B, N)
    out_r.backward(dyl.float())

    nfl = L.tc_effatt_scratch_floats(c, B, N)
    part = torch.empty(nfl, device=dev, dtype=torch.float32)
    out = wide(torch.zeros(B * N, c))
    ctx = torch.empty(B, c, c, device=dev, dtype=torch.float32)
    kstat = torch.empty(B, 2, c, device=dev, dtype=torch.float32)
    dt = wide(torch.zeros(B * N, c))
    g1 = torch.empty(B * N, c, device=dev, dtype=dtype)
    G = {n: torch.zeros_like(P32[n], device=dev, dtype=torch.float32) for n in NAMES}
    f = _lib.TcEffAtt()
    f.t = tl.data_ptr(); f.gamma = Pl["gamma"].data_ptr(); f.beta = Pl["beta"].data_ptr()
    for n in ("wk", "bk", "wq", "bq", "wv", "bv", "wr", "br"):
        setattr(f, n, Pl[n].data_ptr())
    f.out = out.data_ptr(); f.ctx = ctx.data_ptr(); f.kstat = kstat.data_ptr(); f.part = part.data_ptr(); f.part_floats = nfl
    f.dout = dyl.data_ptr(); f.dt = dt.data_ptr(); f.g1 = g1.data_ptr()
    f.dgamma = G["gamma"].data_ptr(); f.dbeta = G["beta"].data_ptr()
    for n in ("wk", "bk", "wq", "bq", "wv", "bv", "wr", "br"):
        setattr(f, "d" + n, G[n].data_ptr())
    f.ldt = f.ldo = f.lddo = f.lddt = c + pad
    f.acc_dt = 0; f.C = c; f.B = B; f.N = N; f.eps = 1e-5
    dt_code = _lib.dtype_code(dtype) if hasattr(_lib, "dtype_code") else {torch.bfloat16: 1, torch.float16: 2}[dtype]
    s = torch.cuda.current_stream().cuda_stream
    L.tc_effatt_fwd(C.byref(f), dt_code, s)
    torch.cuda.synchronize()
    tol = 0.02 if dtype == torch.bfloat16 else 0.005

    def close(a, b, name, k=1.0, scale=None):
        err = float((a.float() - b.float()).abs().max()) / max(float(b.float().abs().max()) if scale is None else scale, 1e-6)
        assert err < tol * k, f"{name}: {err:.4f} (budget {tol * k})"

    close(ctx, ctx_r.detach(), "ctx")
    close(out, out_r.detach(), "out")
    L.tc_effatt_bwd(C.byref(f), dt_code, s)
    torch.cuda.synchronize()
    close(dt, tr.grad, "dt")
    for n in NAMES:
        # d(bk) is zero in exact arithmetic (the softmax over the tokens ignores a per-channel shift of the keys): what is left is
        # the rounding of sum_n dK, measured on the scale of the same sum weighted by the O(1) tokens, dWk
        close(G[n], Pr[n].grad, "d" + n, 1.5, float(Pr["wk"].grad.abs().max()) if n == "bk" else None)
    # accumulate form: dt += ...
    f.acc_dt = 1
    for n in NAMES:
        G[n].zero_()
    L.tc_effatt_bwd(C.byref(f), dt_code, s)
    torch.cuda.synchronize()
    close(dt, 2 * tr.grad, "dt (accumulated)", 1.5)
