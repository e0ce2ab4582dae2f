"""Host time of one hipGraph launch of the captured training step (is the replayed step host-bound?)."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench as B
from transception_amd import MSTransception
from transception_amd.train import FusedSGD, GraphedStep, SegLoss
dev = torch.device("cuda:0")
m = MSTransception(num_classes=9).to(dev).train(); m.set_compute_dtype(torch.bfloat16)
x, y = B.synthetic_batch(16, 224, dev, 1)
opt = FusedSGD(m, lr=0.05, momentum=0.9, weight_decay=1e-4)
for split in (False, True):
    st = GraphedStep(m, SegLoss(9), opt, x, y, None, warmup=2, force_split=split)
    for _ in range(3): st()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): st()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"split={split}: host returns after {1e3 * (t1 - t0) / 10:.2f} ms per step; with the final sync {1e3 * (t2 - t0) / 10:.2f} ms per step")
