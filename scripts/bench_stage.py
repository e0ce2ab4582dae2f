"""The memory-bound stages BASELINE.json's north_star names, alone: RIPM (Patch_Embed_stage of DWConv2d_BN, MSTr.py:704-732) and IFF (CoordAtt,
MSTr.py:1304-1348), forward + backward at the bench shape (B=16, 224^2, bf16), all three encoder stages of each in one replayed hipGraph.

    python scripts/bench_stage.py ripm|iff [--batch 16] [--iters 30]     # prints one JSON line: us per pass, algorithmic bytes, GB/s

Under rocprofv3 (scripts/pmc_stage.sh) the same command gives the kernels' durations and the memory-side bytes (FETCH_SIZE / WRITE_SIZE passes).
Algorithmic bytes: SURVEY.md section 8(d), fused-unit model -- RIPM 0.82 M, IFF 0.55 M activation elements per image forward (inputs read
once, outputs written once), backward = 2 x; weights are negligible here."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transception_amd.model as MM
from transception_amd import MSTransception
from transception_amd.engine import Graph, Var

ELEMS = {"ripm": 0.82e6, "iff": 0.55e6}
DIMS = (64, 128, 320, 512)


def build(which, B, dev, dtype=torch.bfloat16):
    m = MSTransception(num_classes=9).to(dev).train()
    m._ensure_flat(dev)
    m.set_compute_dtype(dtype)
    m._flat_lp = m._flat.to(dtype)
    sides = [56, 28, 14, 7]
    ins, gys = [], []
    for s in (1, 2, 3):                                   # encoder stages 2, 3, 4
        side_in, C = sides[s - 1], DIMS[s - 1]
        so = sides[s]
        if which == "ripm":
            ins.append(torch.randn(B * side_in * side_in, C, device=dev).to(dtype))
            gys.append(torch.randn(3 * B * so * so, C, device=dev).to(dtype))
        else:
            ins.append(torch.randn(B * so * so, 4 * C, device=dev).to(dtype))
            gys.append(torch.randn(B * so * so, DIMS[s], device=dev).to(dtype))

    def one_pass():
        m._used_views = {}
        G = Graph(dtype, dev, training=True, record=True)
        outs = []
        for s in (1, 2, 3):
            x = Var(ins[s - 1], requires_grad=True)
            if which == "ripm":
                o, _ = MM._ripm(m, G, x, f"backbone.patch_embed_stage{s + 1}", B, sides[s - 1])
            else:
                so = sides[s]
                o = MM._coord_att(m, G, x, f"backbone.mhca_stage{s + 1}.aggregate", B, so, G.new(B * so * so, DIMS[s]))
            outs.append(o)
        for o, g in zip(outs, gys):
            o.root.grad_t = g
            o.root.whole_written = True
        G.backward()
        return G.n_launch
    return m, one_pass


def main():
    which = sys.argv[1]
    B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 16
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 30
    dev = torch.device("cuda:0")
    m, one_pass = build(which, B, dev)
    for _ in range(3):
        one_pass()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        one_pass()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            one_pass()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    by = 3.0 * ELEMS[which] * B * 2
    print(json.dumps({"stage": which, "batch": B, "us_per_fwd_bwd": us, "algorithmic_bytes": by, "achieved_GBps": by / us / 1e3,
                      "frac_of_8TBps": by / us / 1e3 / 8000.0}))


if __name__ == "__main__":
    main()
