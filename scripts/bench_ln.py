"""LayerNorm forward / backward micro-benchmark (graph replay of 20 launches): python scripts/bench_ln.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transception_amd.engine import Graph, P, Var

dev = torch.device("cuda:0")
dt = torch.bfloat16


def replay_us(fn, n=20):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for rows, C in ((97216, 64), (50176, 64), (37632, 64), (12544, 128), (9408, 128), (50176, 256), (9408, 512)):
    x = torch.randn(rows, C, device=dev).to(dt)
    gam, bet = (torch.randn(C, device=dev).to(dt) for _ in range(2))
    gg, gb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    gy = torch.randn(rows, C, device=dev).to(dt)

    def fwd():
        G = Graph(dt, dev, True, False)
        G.layernorm(Var(x), P(gam, None), P(bet, None))

    G = Graph(dt, dev, True, True)
    xv = Var(x)
    out = G.layernorm(xv, P(gam, gg), P(bet, gb))
    out.root.grad_t = gy; out.root.whole_written = True
    tape = list(G.tape)

    def bwd():
        xv.root.grad_t = None; xv.root.whole_written = False; xv.root.written = []
        G.tape = list(tape)
        G.backward()
    by = rows * C * 2
    tf, tb = replay_us(fwd), replay_us(bwd)
    print(f"rows {rows:6d} C {C:4d}: fwd {tf:6.1f} us ({2 * by / tf / 1e3:6.0f} GB/s)   bwd {tb:6.1f} us ({3 * by / tb / 1e3:6.0f} GB/s)")
