"""LayerNorm microbenchmark (bf16) at the model's shapes: fwd, bwd-dx, bwd-params vs a same-size copy."""
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
def run(rows, C, act=0):
    x = torch.randn(rows, C, device=dev).bfloat16(); y = torch.empty_like(x); dy = torch.randn(rows, C, device=dev).bfloat16(); dx = torch.empty_like(x)
    g = torch.ones(C, device=dev).bfloat16(); b = torch.zeros(C, device=dev).bfloat16(); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
    f = t(lambda: L.tc_layernorm_fwd(x.data_ptr(), C, g.data_ptr(), b.data_ptr(), y.data_ptr(), C, mean.data_ptr(), rstd.data_ptr(), rows, C, 1e-5, act, 1, 0, TC_BF16, st))
    d = t(lambda: L.tc_layernorm_bwd(dy.data_ptr(), C, x.data_ptr(), C, g.data_ptr(), b.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), C, None, 0, None, None, rows, C, act, 1, 0, None, 0, TC_BF16, st))
    ns = L.tc_layernorm_bwd_scratch_floats(rows, C, 1); sc = torch.zeros(ns, device=dev)
    fz = t(lambda: L.tc_layernorm_bwd(dy.data_ptr(), C, x.data_ptr(), C, g.data_ptr(), b.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), C, None, 0, dg.data_ptr(), db.data_ptr(), rows, C, act, 1, 0, sc.data_ptr(), ns, TC_BF16, st))
    p = t(lambda: L.tc_layernorm_bwd_params(dy.data_ptr(), C, x.data_ptr(), C, g.data_ptr(), b.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dg.data_ptr(), db.data_ptr(), rows, C, act, 1, 0, TC_BF16, st))
    c = t(lambda: y.copy_(x))
    print(f"rows={rows:6d} C={C:5d} act={act}: fwd {f:6.1f}  dx {d:6.1f}  params {p:6.1f}  fused {fz:6.1f}  copy {c:6.1f} us")
run(50176, 64); run(50176, 256, 4); run(12544, 128); run(12544, 512, 4); run(3136, 320); run(3136, 1280, 4); run(784, 512); run(784, 2048, 4); run(97216, 64); run(150528, 64)
