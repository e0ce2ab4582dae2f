"""Census of the GEMMs of one training step: every tc_gemm call of an eager, single-stream step is replayed alone
(20 timed repeats on the same pointers, accumulate forced so nothing is clobbered beyond what the step already tolerates)
and reported by shape class with its achieved TFLOP/s and the HBM-floor time.    python scripts/gemm_census.py"""
import os, sys, collections, ctypes as C
os.environ["TC_STREAMS"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import transception_amd.engine as engine
from transception_amd import MSTransception
from transception_amd.seeded_init import seeded_state_dict
from transception_amd.train import FusedSGD, SegLoss, train_step
from transception_amd._lib import TcGemm

dev = torch.device("cuda", 0)
model = MSTransception(num_classes=9); model.load_state_dict(seeded_state_dict(), strict=True); model.to(dev).train()
model.set_compute_dtype(torch.bfloat16); model._ensure_flat(dev)
loss_fn = SegLoss(9); opt = FusedSGD(model, lr=0.05, momentum=0.9, weight_decay=1e-4)
g = torch.Generator().manual_seed(1)
x = ((torch.rand(16, 1, 224, 224, generator=g) - 0.5) / 0.5).to(dev); y = torch.randint(0, 9, (16, 224, 224), generator=g).to(dev)
for _ in range(2): train_step(model, loss_fn, opt, x, y, None)
torch.cuda.synchronize()

calls = []
orig = engine.Graph._gemm
def rec(self, *a, **k):
    orig(self, *a, **k)
    A, lda, B, ldb, Cm, ldc, M, N, K, tA, tB = a[:11]
    gs = TcGemm(A, B, Cm, k.get('bias'), k.get('R'), M, N, K, lda, ldb, ldc, k.get('ldr', 0), tA, tB, k.get('nb1', 1), k.get('nb2', 1),
                *k.get('sA', (0, 0)), *k.get('sB', (0, 0)), *k.get('sC', (0, 0)), *k.get('sR', (0, 0)), k.get('alpha', 1.0), k.get('acc', 0),
                k.get('act', 0), k.get('splitk', 1), self.dt, k.get('c_f32', 0), k.get('atomic', 0), k.get('rowsum'), k.get('sbias', 0), k.get('srow', 0), 0, 0)
    calls.append((gs, self.stream))
engine.Graph._gemm = rec
# keep every temporary alive so the recorded pointers stay valid: hold the graph
keep = []
orig_bwd = model._backward if hasattr(model, "_backward") else None
train_step(model, loss_fn, opt, x, y, None)
torch.cuda.synchronize()
engine.Graph._gemm = orig
L = engine.lib() if hasattr(engine, "lib") else None
from transception_amd._lib import lib
L = lib()
res = collections.defaultdict(lambda: [0.0, 0, 0.0, 0.0])
s = torch.cuda.current_stream()
for gs, st in calls:
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    for _ in range(3): L.tc_gemm(C.byref(gs), s.cuda_stream)
    e0.record(s)
    for _ in range(20): L.tc_gemm(C.byref(gs), s.cuda_stream)
    e1.record(s); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    nb = gs.nb1 * gs.nb2
    key = (gs.M, gs.N, gs.K, gs.transA, gs.transB, nb, gs.splitk, gs.c_f32, gs.atomic, 1 if gs.rowsum else 0)
    fl = 2.0 * gs.M * gs.N * gs.K * nb
    by = nb * (gs.M * gs.K * 2 + gs.N * gs.K * 2 + gs.M * gs.N * (4 if gs.c_f32 else 2))
    r = res[key]; r[0] += us; r[1] += 1; r[2] = fl; r[3] = by
tot = sum(r[0] for r in res.values())
print(f"{len(calls)} GEMM calls, {len(res)} shape classes, isolated total {tot/1e3:.2f} ms")
print("   total_us  n   avg_us  TF/s  hbm_floor_us   (M, N, K, tA, tB, nb, splitk, c_f32, atomic, rowsum)")
for k, r in sorted(res.items(), key=lambda kv: -kv[1][0])[:70]:
    avg = r[0] / r[1]
    print(f"  {r[0]:8.1f} {r[1]:3d} {avg:8.1f} {r[2]/avg/1e6:6.1f} {r[3]/8e6:8.2f}   {k}")
