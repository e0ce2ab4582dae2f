"""A few hundred captured training steps on the synthetic Synapse set (bf16 storage, loader-fed, the default kernels incl. the
hand-scheduled attention stream): the loss must fall and stay finite.  python scripts/train_sanity.py [steps]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transception_amd import MSTransception, data as D
from transception_amd.seeded_init import seeded_state_dict
from transception_amd.train import FusedSGD, GraphedStep, SegLoss, cosine_lr
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
tmp = tempfile.mkdtemp()
D.write_synthetic_synapse(tmp + "/a", tmp + "/l", n_cases=4, slices_per_case=32, size=512, seed=1)
ds = D.SynapseSlices(tmp + "/a", tmp + "/l")
model = MSTransception(num_classes=9); model.load_state_dict(seeded_state_dict(), strict=True); model.to(dev).train()
model.set_compute_dtype(torch.bfloat16); model._ensure_flat(dev)
loss_fn, opt = SegLoss(9), FusedSGD(model, lr=0.05, momentum=0.9, weight_decay=1e-4)
loader = D.DeviceLoader(ds, 16, device=dev, epochs=steps * 16 // len(ds) + 2, readers=8)
x0 = torch.zeros(16, 1, 224, 224, device=dev); y0 = torch.zeros(16, 224, 224, dtype=torch.int64, device=dev)
fed, hist = {}, []
t0 = time.perf_counter()
for i, slot in enumerate(loader.iter_raw()):
    if i == steps:
        break
    st = fed.get(slot["index"])
    if st is None:
        st = fed[slot["index"]] = GraphedStep(model, loss_fn, opt, x0, y0, None, warmup=1, pre=loader.slot_preprocess(slot))
    loss, ce, dice = st()
    opt.set_lr(cosine_lr(0.05, i + 1, steps))
    if i % 25 == 0 or i == steps - 1:
        hist.append((i, float(loss), float(ce), float(dice)))
        print(f"step {i:4d} loss {hist[-1][1]:.4f} ce {hist[-1][2]:.4f} dice {hist[-1][3]:.4f}", flush=True)
torch.cuda.synchronize()
print(f"{steps} steps in {time.perf_counter() - t0:.1f} s; finite: {all(map(lambda h: h[1] == h[1] and abs(h[1]) < 1e4, hist))}; "
      f"loss {hist[0][1]:.4f} -> {hist[-1][1]:.4f}")
assert hist[-1][1] < 0.8 * hist[0][1]
