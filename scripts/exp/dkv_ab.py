"""A/B of the hand-scheduled dK/dV stream against the compiler-scheduled kernel on the bench shape: same inputs, both libraries' outputs."""
import os, sys, subprocess, json
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "worker":
    import torch, ctypes as C
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from transception_amd.engine import Graph, Var
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B, nq, Nk = (int(a) for a in sys.argv[2:5]) if len(sys.argv) > 4 else (16, 0, 784)
    nqs = [3136, 1568, 980, 392] if nq == 0 else [nq]
    rows = B * sum(nqs)
    G = Graph(torch.bfloat16, dev, training=True, record=True)
    q = Var((torch.randn(rows, 64, device=dev) * 0.5).bfloat16()); k = Var((torch.randn(B * Nk, 64, device=dev)).bfloat16()); v = Var(torch.randn(B * Nk, 64, device=dev).bfloat16())
    o = G.attention_seg(q, k, v, B, nqs, Nk, 0.125, q_prescaled=True)
    o.root.grad_t = torch.randn(rows, 64, device=dev).bfloat16(); o.root.whole_written = True
    G.backward(); torch.cuda.synchronize()
    np.savez(sys.argv[5] if len(sys.argv) > 5 else "/tmp/dkv_ab.npz", dq=G.grad_of(q).float().cpu().numpy(), dk=G.grad_of(k).float().cpu().numpy(), dv=G.grad_of(v).float().cpu().numpy())
    sys.exit(0)
for shape in ([16, 0, 784], [3, 100, 70], [2, 6076, 33], [1, 64, 64], [5, 97, 500]):
    res = {}
    for asm in ("1", "0"):
        out = f"/tmp/dkv_ab_{asm}.npz"
        subprocess.run([sys.executable, __file__, "worker", *map(str, shape), out], check=True, env=dict(os.environ, TC_ATTN_DKV_ASM=asm, TC_ATTN_DQ_ASM=asm), timeout=120)
        res[asm] = np.load(out)
    line = []
    for kname in ("dq", "dk", "dv"):
        a, b = res["1"][kname], res["0"][kname]
        line.append(f"{kname}: max|d| {np.abs(a - b).max():.3e} ref max {np.abs(b).max():.3e} nan {int(np.isnan(a).sum())}")
    print(shape, " | ".join(line))
