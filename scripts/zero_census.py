"""Which gradient buffers of one training step ask the engine's zero arena for a zeroed start (engine._ZeroArena), and which torch fills run:
python scripts/zero_census.py"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import transception_amd.engine as E
from transception_amd import MSTransception
from transception_amd.seeded_init import seeded_state_dict
from transception_amd.train import FusedSGD, SegLoss, train_step
dev = torch.device("cuda:0")
m = MSTransception(num_classes=9); m.load_state_dict(seeded_state_dict(), strict=True); m.to(dev).train(); m.set_compute_dtype(torch.bfloat16)
opt = FusedSGD(m, lr=0.05); loss = SegLoss(9)
x = torch.rand(16, 1, 224, 224, device=dev); y = torch.randint(0, 9, (16, 224, 224), device=dev)
train_step(m, loss, opt, x, y)
req = collections.Counter()
orig = E._ZeroArena.zeros_like
def spy(self, t):
    fr = [f for f in traceback.extract_stack()[:-1] if "engine.py" in f.filename or "model.py" in f.filename][-4:]
    req[(tuple(t.shape), " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in fr))] += t.numel() * t.element_size()
    return orig(self, t)
E._ZeroArena.zeros_like = spy
tz, tzl = torch.zeros, torch.zeros_like
big = collections.Counter()
def z(*a, **k):
    r = tz(*a, **k)
    if r.numel() * r.element_size() > 1 << 20:
        fr = traceback.extract_stack()[-2]
        big[(tuple(r.shape), str(r.dtype), f"{os.path.basename(fr.filename)}:{fr.lineno}")] += r.numel() * r.element_size()
    return r
torch.zeros = z
train_step(m, loss, opt, x, y)
torch.cuda.synchronize()
print("zero-arena requests (MB):")
for k, v in sorted(req.items(), key=lambda kv: -kv[1])[:25]:
    print(f"  {v / 1e6:8.1f}  {k[0]}  {k[1]}")
print("total", sum(req.values()) / 1e6, "MB")
print("torch.zeros > 1 MB:")
for k, v in sorted(big.items(), key=lambda kv: -kv[1]):
    print(f"  {v / 1e6:8.1f}  {k}")
