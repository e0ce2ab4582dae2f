import sys, time, torch
sys.path.insert(0, "/root/repo")
from transception_amd import MSTransception
from transception_amd.seeded_init import seeded_state_dict
from transception_amd.train import FusedSGD, GraphedStep, SegLoss
from bench import synthetic_batch
dev = torch.device("cuda:0")
m = MSTransception(9); m.load_state_dict(seeded_state_dict()); m.to(dev).train(); m.set_compute_dtype(torch.bfloat16); m._ensure_flat(dev)
x, y = synthetic_batch(16, 224, dev, 1)
opt = FusedSGD(m); lf = SegLoss(9)
step = GraphedStep(m, lf, opt, x, y, None, warmup=2)
for _ in range(3): step()
torch.cuda.synchronize()
for n in (1, 5, 20):
    t0 = time.perf_counter()
    for _ in range(n): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"n={n}: host enqueue {1e3*(t1-t0)/n:.2f} ms/step, total {1e3*(t2-t0)/n:.2f} ms/step")
# GPU-side time of one replay with events
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record(); step(); e1.record(); torch.cuda.synchronize()
print("single replay GPU time (events):", e0.elapsed_time(e1), "ms")
