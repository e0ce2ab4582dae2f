"""Which gradient buffers of one training step are carved from the zero-filled arena (engine._Zeros), and from where."""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import transception_amd.engine as E
from transception_amd import MSTransception
from transception_amd.seeded_init import seeded_state_dict, seeded_input, seeded_labels
from transception_amd.train import FusedSGD, SegLoss, train_step
dev = torch.device("cuda:0")
m = MSTransception(num_classes=9); m.load_state_dict(seeded_state_dict(), strict=True); m.to(dev).train(); m.set_compute_dtype(torch.bfloat16)
cls = [c for c in vars(E).values() if isinstance(c, type) and hasattr(c, "zeros_like") and hasattr(c, "close")][0]
orig = cls.zeros_like
sites = collections.OrderedDict()
def logged(self, t):
    st = traceback.extract_stack(limit=8)
    key = " <- ".join(f"{f.name}:{f.lineno}" for f in reversed(st[:-1]) if "engine.py" in f.filename or "model.py" in f.filename)[:150]
    sites.setdefault(key, [0, 0]); sites[key][0] += 1; sites[key][1] += t.numel() * t.element_size()
    return orig(self, t)
cls.zeros_like = logged
x = torch.rand(16, 1, 224, 224, device=dev) * 2 - 1
y = torch.randint(0, 9, (16, 224, 224), device=dev)
loss_fn, opt = SegLoss(9), FusedSGD(m, lr=0.05, momentum=0.9, weight_decay=1e-4)
train_step(m, loss_fn, opt, x, y, None)
sites.clear()
train_step(m, loss_fn, opt, x, y, None)
torch.cuda.synchronize()
tot = 0
for k, (n, b) in sorted(sites.items(), key=lambda kv: -kv[1][1]):
    print(f"{b / 1e6:9.2f} MB  {n:3d}x  {k}")
    tot += b
print(f"total {tot / 1e6:.1f} MB")
