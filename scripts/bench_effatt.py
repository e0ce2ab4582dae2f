"""EfficientAttention (MSTr.py:106-143) + its LayerNorm and residual, forward + backward, one site shape, through the engine:
python scripts/bench_effatt.py [C B N] [--reps 20]   (run under rocprofv3 --kernel-trace --stats for the per-kernel numbers)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
a = [x for x in sys.argv[1:] if not x.startswith("--")]
C, B, N = (int(v) for v in a[:3]) if len(a) >= 3 else (64, 16, 3136)
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 20
import transception_amd.model as MM
from transception_amd.engine import Graph, P, Var
dev, dtype = torch.device("cuda:0"), torch.bfloat16
g = torch.Generator().manual_seed(1)


from transception_amd import MSTransception
from transception_amd.seeded_init import seeded_state_dict
M = MSTransception(num_classes=9); M.load_state_dict(seeded_state_dict(), strict=True); M.to(dev).train(); M.set_compute_dtype(dtype)
with torch.no_grad():
    M(torch.rand(2, 1, 224, 224, device=dev))                      # builds the flat arenas and the 16-bit working copy
M._used_views = {}
BLK = {64: "backbone.block1.0", 128: "decoder_1.layer_former_1", 320: "decoder_2.layer_former_1"}[C]
x = torch.randn(B * N, C, generator=g).to(dev).to(dtype)
gout = torch.randn(B * N, C, generator=g).to(dev).to(dtype)


def run():
    G = Graph(dtype, dev, training=True, record=True)
    t = Var(x)
    out = MM._eff_attention(M, G, MM._ln(M, G, t, BLK + ".norm1"), BLK + ".attn", B, N, residual=t)[0]
    out.root.grad_t = gout
    out.root.whole_written = True
    G.backward()
    return G.n_launch
for _ in range(3):
    nl = run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record(); torch.cuda.synchronize()
print(f"eff_attention C={C} B={B} N={N}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per fwd+bwd (eager), {nl} forward launches recorded")
