"""Eager step with every fresh engine buffer NaN-filled (TC_DEBUG_POISON=1): which parameter gradients / outputs read unwritten memory?"""
import os, sys, torch
os.environ["TC_DEBUG_POISON"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transception_amd import MSTransception
from transception_amd.seeded_init import seeded_state_dict
from transception_amd.train import FusedSGD, SegLoss, train_step
dev = torch.device("cuda", 0)
model = MSTransception(num_classes=9); model.load_state_dict(seeded_state_dict(), strict=True); model.to(dev).train()
model.set_compute_dtype(torch.bfloat16); model._ensure_flat(dev)
loss_fn = SegLoss(9); opt = FusedSGD(model, lr=0.0, momentum=0.9, weight_decay=0.0)
g = torch.Generator().manual_seed(1234)
x = ((torch.rand(16, 1, 224, 224, generator=g) - 0.5) / 0.5).to(dev); y = torch.randint(0, 9, (16, 224, 224), generator=g).to(dev)
loss, ce, dice = train_step(model, loss_fn, opt, x, y, None)
torch.cuda.synchronize()
print("loss", float(loss))
bad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
print(len(bad), "params with non-finite grads; first:", bad[:12])
