"""Host-side time per iteration of the loader-fed training loop (is the host able to stay ahead of the GPU?)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transception_amd import MSTransception, data as D
from transception_amd.seeded_init import seeded_state_dict
from transception_amd.train import FusedSGD, GraphedStep, SegLoss
dev = torch.device("cuda:0")
tmp = tempfile.mkdtemp()
D.write_synthetic_synapse(tmp + "/a", tmp + "/l", n_cases=4, slices_per_case=32, size=512, seed=1)
ds = D.SynapseSlices(tmp + "/a", tmp + "/l")
model = MSTransception(num_classes=9); model.load_state_dict(seeded_state_dict(), strict=True); model.to(dev).train()
model.set_compute_dtype(torch.bfloat16); model._ensure_flat(dev)
loss_fn = SegLoss(9); opt = FusedSGD(model, lr=0.05)
loader = D.DeviceLoader(ds, 16, device=dev, epochs=20, readers=8)
it = iter(loader)
x, y = next(it)
step = GraphedStep(model, loss_fn, opt, x, y, None, warmup=2)
tn, ts = [], []
torch.cuda.synchronize()
t0 = time.perf_counter()
evs, ahead = [], []
for i in range(60):
    a = time.perf_counter(); x, y = next(it); b = time.perf_counter(); step(x, y); c = time.perf_counter()
    tn.append(b - a); ts.append(c - b)
    e = torch.cuda.Event(); e.record(); evs.append(e)
    ahead.append(sum(0 if ev.query() else 1 for ev in evs[-12:]))
torch.cuda.synchronize()
tot = time.perf_counter() - t0
import statistics as st
print(f"60 steps in {tot*1e3:.1f} ms = {tot/60*1e3:.2f} ms/step; host next(): median {st.median(tn)*1e3:.2f} max {max(tn)*1e3:.2f} ms; host step(): median {st.median(ts)*1e3:.2f} max {max(ts)*1e3:.2f} ms")
print("steps queued ahead of the GPU after each launch:", " ".join(str(v) for v in ahead[:40]))
print("next() ms:", " ".join(f"{v*1e3:.1f}" for v in tn[:40]))
print("step() ms:", " ".join(f"{v*1e3:.1f}" for v in ts[:40]))
