"""Phase cycles of ln_bwd_kernel over one training step (library built with -DTC_LN_TIMING): per kind
(TA, TB, fp32 out, FFN hook) the workgroup count and the mean cycles in setup / K loop / epilogue.
usage: TC_LIB_PATH=scripts/exp/libtc_gemmtiming.so python scripts/exp/gemm_timing.py"""
import ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from transception_amd import MSTransception
from transception_amd.seeded_init import seeded_state_dict, seeded_input, seeded_labels
from transception_amd.train import FusedSGD, SegLoss, train_step
dev = torch.device("cuda:0")
m = MSTransception(num_classes=9)
m.load_state_dict(seeded_state_dict(), strict=True)
m.to(dev).train()
m.set_compute_dtype(torch.bfloat16)
x = torch.from_numpy(seeded_input(16)).to(dev); y = torch.from_numpy(seeded_labels(16)).to(dev)
loss_fn, opt = SegLoss(9), FusedSGD(m, lr=0.01, momentum=0.9, weight_decay=1e-4)
for _ in range(2):
    train_step(m, loss_fn, opt, x, y)
torch.cuda.synchronize()
L = C.CDLL(os.environ["TC_LIB_PATH"])
L.tc_ln_dbg_read.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros(1024 * 8, dtype=np.uint64)
L.tc_ln_dbg_read(buf.ctypes.data, 1)
train_step(m, loss_fn, opt, x, y)
torch.cuda.synchronize()
L.tc_ln_dbg_read(buf.ctypes.data, 0)
t = buf.reshape(1024, 8).astype(np.float64)
t = t[t[:, 2] + t[:, 3] > 0]
print(len(t), "sampled ln_bwd workgroups (every 61st)")
print("GS  C   rows     samples   params   row loop   LDS partials   park+arrive   fold(last arrivers only)")
import collections
groups = collections.defaultdict(list)
for r in t:
    groups[(int(r[0]) % 1000, int(r[0]) // 1000, int(r[1]))].append(r)
for key, rows_ in sorted(groups.items(), key=lambda kv: -len(kv[1])):
    a = np.array(rows_)
    last = a[a[:, 6] > 0]
    print(f"{key[0]:3d} {key[1]:4d} {key[2]:7d}  {len(rows_):6d}   {a[:,2].mean():7.0f} {a[:,3].mean():9.0f} {a[:,4].mean():10.0f} {a[:,5].mean():12.0f}   {last[:,6].mean() if len(last) else 0:8.0f} ({len(last)})")
