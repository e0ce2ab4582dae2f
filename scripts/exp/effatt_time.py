"""Time the fused EfficientAttention launches alone (B = 16, N = 3136, C = 64, bf16): run under
`rocprofv3 --kernel-trace --stats` for the per-kernel durations."""
import ctypes as C
import sys
import torch
from transception_amd import _lib

L = _lib.lib()
dev = torch.device("cuda:0")
B, N, c = 16, int(sys.argv[1]) if len(sys.argv) > 1 else 3136, 64
dt = torch.bfloat16
P = {n: (torch.randn((c, c) if n.startswith("w") else (c,), device=dev) * 0.2).to(dt) for n in ["gamma", "beta", "wk", "bk", "wq", "bq", "wv", "bv", "wr", "br"]}
t = torch.randn(B * N, c, device=dev).to(dt); dy = torch.randn(B * N, c, device=dev).to(dt)
nfl = L.tc_effatt_scratch_floats(c, B, N)
part = torch.empty(nfl, device=dev); out = torch.empty_like(t); ctx = torch.empty(B, c, c, device=dev); ks = torch.empty(B, 2, c, device=dev)
dtt = torch.empty_like(t); g1 = torch.empty_like(t)
G = {n: torch.zeros(P[n].shape, device=dev) for n in P}
f = _lib.TcEffAtt()
f.t = t.data_ptr()
for n in P:
    setattr(f, n, P[n].data_ptr()); setattr(f, "d" + n, G[n].data_ptr())
f.out = out.data_ptr(); f.ctx = ctx.data_ptr(); f.kstat = ks.data_ptr(); f.part = part.data_ptr(); f.part_floats = nfl
f.dout = dy.data_ptr(); f.dt = dtt.data_ptr(); f.g1 = g1.data_ptr()
f.ldt = f.ldo = f.lddo = f.lddt = c; f.acc_dt = 0; f.C = c; f.B = B; f.N = N; f.eps = 1e-5
s = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    L.tc_effatt_fwd(C.byref(f), 1, s); L.tc_effatt_bwd(C.byref(f), 1, s)
torch.cuda.synchronize()
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
R = 50
e0.record()
for _ in range(R):
    L.tc_effatt_fwd(C.byref(f), 1, s)
e1.record()
for _ in range(R):
    L.tc_effatt_bwd(C.byref(f), 1, s)
e2.record()
torch.cuda.synchronize()
print(f"N={N}: fwd {e0.elapsed_time(e1) / R * 1e3:.1f} us   bwd {e1.elapsed_time(e2) / R * 1e3:.1f} us")
