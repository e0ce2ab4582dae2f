"""Every GEMM-family C-ABI call of one eager training step (tc_gemm / tc_gemm_pair / tc_gemm_multi) replayed alone and timed:
shape, achieved TFLOP/s and bytes/s against the algorithmic traffic.   python scripts/gemm_census2.py"""
import os, sys, collections, ctypes as C, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transception_amd import MSTransception
from transception_amd.seeded_init import seeded_state_dict
from transception_amd.train import FusedSGD, SegLoss, train_step
from transception_amd._lib import TcGemm, lib

dev = torch.device("cuda", 0)
model = MSTransception(num_classes=9); model.load_state_dict(seeded_state_dict(), strict=True); model.to(dev).train()
model.set_compute_dtype(torch.bfloat16); model._ensure_flat(dev)
loss_fn = SegLoss(9); opt = FusedSGD(model, lr=0.05)
g = torch.Generator().manual_seed(1)
x = ((torch.rand(16, 1, 224, 224, generator=g) - 0.5) / 0.5).to(dev); y = torch.randint(0, 9, (16, 224, 224), generator=g).to(dev)
for _ in range(2): train_step(model, loss_fn, opt, x, y, None)
torch.cuda.synchronize()
L = lib()
calls = []
def clone(gs):
    c = TcGemm(); C.memmove(C.byref(c), C.byref(gs), C.sizeof(TcGemm)); return c
o1, o2, o3 = L.tc_gemm, L.tc_gemm_pair, L.tc_gemm_multi
def r1(gp, st): calls.append(("gemm", [clone(gp._obj)])); o1(gp, st)
def r2(a, b, st): calls.append(("pair", [clone(a._obj), clone(b._obj)])); o2(a, b, st)
def r3(arr, n, st): calls.append(("multi", [clone(arr[i]) for i in range(n)])); o3(arr, n, st)
L.tc_gemm, L.tc_gemm_pair, L.tc_gemm_multi = r1, r2, r3
train_step(model, loss_fn, opt, x, y, None)
torch.cuda.synchronize()
L.tc_gemm, L.tc_gemm_pair, L.tc_gemm_multi = o1, o2, o3
s = torch.cuda.current_stream(); st = s.cuda_stream
def work(gs):
    nb = gs.nb1 * gs.nb2
    fl = 2.0 * gs.M * gs.N * gs.K * nb
    by = nb * (gs.M * gs.K * 2 + gs.N * gs.K * 2 + gs.M * gs.N * (4 if gs.c_f32 else 2))
    return fl, by
rows = []
for kind, gl in calls:
    for q in gl: q.accumulate = 1 if q.c_f32 else q.accumulate
    if kind == "gemm": fn = lambda: o1(C.byref(gl[0]), st)
    elif kind == "pair": fn = lambda: o2(C.byref(gl[0]), C.byref(gl[1]), st)
    else:
        arr = (TcGemm * len(gl))(*gl); fn = lambda: o3(arr, len(gl), st)
    for _ in range(2): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(10): fn()
    e1.record(s); e1.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = sum(work(q)[0] for q in gl); by = sum(work(q)[1] for q in gl)
    desc = " + ".join(f"{q.M}x{q.N}x{q.K}{'T' if q.transA else ''}{'' if q.transB else 'n'}{'*%d' % (q.nb1*q.nb2) if q.nb1*q.nb2 > 1 else ''}" for q in gl[:4]) + (" ..." if len(gl) > 4 else "")
    rows.append((kind, desc, us, fl, by))
agg = collections.OrderedDict()
for kind, desc, us, fl, by in rows:
    a = agg.setdefault((kind, desc), [0, 0.0, fl, by]); a[0] += 1; a[1] += us
tot = sum(r[2] for r in rows)
print(f"{len(rows)} GEMM-family launches, {tot/1e3:.2f} ms when replayed alone")
floor = 0.0
for (kind, desc), (n, us, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    t_floor = max(by / 4.5e6, fl / 1.2e9)             # us at 4.5 TB/s or 1.2 PFLOP/s
    print(f"{us/1e3:6.2f} ms {n:3d}x avg {us/n:6.1f} us  floor {t_floor:5.1f} us  {fl/(us/n)/1e6:6.1f} TF/s {by/(us/n)/1e6:5.2f} TB/s  {kind:5s} {desc}")
print("sum of floors (4.5 TB/s | 1.2 PF/s, +3 us launch each): %.2f ms" % (sum(max(r[4] / 4.5e6, r[3] / 1.2e9) + 3.0 for r in rows) / 1e3))
