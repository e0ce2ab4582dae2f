"""MixFFN_skip site benchmark: fused (engine.Graph.mixffn) vs unfused composition, forward + backward, one site shape.

    python scripts/bench_ffn.py [C B H W groups] [--reps 5] [--only fused|unfused]
Run under rocprofv3 (--kernel-trace --stats, or --pmc ...) to see the per-kernel numbers of a single site."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_ffn_fused_gpu import _params  # noqa: E402


def main():
    a = [x for x in sys.argv[1:] if not x.startswith("--")]
    C, B, H, W, groups = (int(v) for v in a[:5]) if len(a) >= 5 else (64, 16, 56, 56, 1)
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 5
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    import transception_amd.model as MM
    from transception_amd.engine import Graph, P, Var
    dev = torch.device("cuda:0")
    dtype = torch.bfloat16
    gen = torch.Generator().manual_seed(1)
    flat, offs, shapes, tot = _params(C, groups, gen)
    rows = groups * B * H * W
    pf = flat.to(dev); pl = pf.to(dtype); gf = torch.zeros_like(pf)
    x = torch.randn(rows, C, generator=gen).to(dev).to(dtype)
    gout = torch.randn(rows, C, generator=gen).to(dev).to(dtype)

    def mk(k):
        n = int(torch.tensor(shapes[k]).prod())
        return P(pl[offs[k]:offs[k] + n].view(shapes[k]), gf[offs[k]:offs[k] + n].view(shapes[k]), tot if groups > 1 else 0)

    def run(fused):
        G = Graph(dtype, dev, training=True, record=True)
        xv = Var(x)
        ctx = G.grouped(groups, tot) if groups > 1 else None
        if ctx:
            ctx.__enter__()
        W1, b1, wd, bd, lg, lb, W2, b2 = (mk(k) for k in ("W1", "b1", "wd", "bd", "lg", "lb", "W2", "b2"))
        if fused:
            out = G.mixffn([dict(x=xv, fc1=(W1, b1), dw=(wd, bd), ln=(lg, lb), fc2=(W2, b2), geo=(B, H, W), residual=None)])[0]
        else:
            h = G.linear(xv, W1, b1)
            d = G.dwconv(h, wd, bd, B, H, W, 3, 1, True)
            a_ = G.layernorm(d, lg, lb, 1e-5, MM.ACT_GELU)
            out = G.linear(a_, W2, b2)
        out.root.grad_t = gout
        out.root.whole_written = True
        G.backward()
        if ctx:
            ctx.__exit__(None, None, None)
    for fused in (True, False):
        if only and (only == "fused") != fused:
            continue
        for _ in range(2):
            run(fused)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run(fused)
        e1.record()
        torch.cuda.synchronize()
        print(f"{'fused' if fused else 'unfused'}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per fwd+bwd (eager launches, C={C} B={B} {H}x{W} groups={groups})")


if __name__ == "__main__":
    main()
