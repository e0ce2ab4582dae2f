import sys, math, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_model_gpu as T
from golden_util import load
import transception_amd.model as MM
from transception_amd.seeded_init import seeded_tensor
model = T._fresh() if hasattr(T, "_fresh") else None
gold = load("modules.npz")
tag, B = "resblock_s3", 2
res = {}
for dtype in (torch.float32, torch.bfloat16):
    x = torch.from_numpy(seeded_tensor(f"{tag}/x0", (B, 128, 14, 14)))
    G = T._graph(model, dtype=dtype); model._gflat.zero_()
    v = T._var(T._tok(x), dtype=dtype)
    out = MM._resblock(model, G, v, "backbone.mhca_stage3.InvRes", B, 14, G.new(B * 196, 128))
    y = T._untok(out.data.float().cpu(), B, 14, 14)
    g = torch.from_numpy(seeded_tensor(f"{tag}/g", tuple(y.shape)))
    T._finish(G, [out], [T._tok(g).to(dtype)])
    res[dtype] = {}
    for k in [k[len(tag) + 4:-6] for k in gold.files if k.startswith(f"{tag}/gw/") and k.endswith("/shape")]:
        off, shape = model._index[k]
        res[dtype][k] = model._gflat[off:off + math.prod(shape)].view(shape).float().cpu().clone()
model.set_compute_dtype(torch.float32)
for k, a in res[torch.float32].items():
    b = res[torch.bfloat16][k]
    print(f"{k:60s} |fp32| max {a.abs().max():9.3e} norm {a.norm():9.3e}  bf16-fp32: max {(a-b).abs().max():9.3e} relL2 {((a-b).norm()/a.norm()).item():.4f}")
