import ctypes as C, sys, torch
sys.path.insert(0, "/root/repo")
from transception_amd._lib import lib, TC_BF16
L = lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream; d = 64
def run(q, k, v, B, nq, Nk):
    rows = q.shape[0]
    kv = torch.cat([k, v], 1).contiguous()
    o = torch.zeros(rows, d, device=dev).bfloat16(); lse = torch.zeros(rows, device=dev)
    nqc = (C.c_int * 4)(*(list(nq) + [0] * (4 - len(nq))))
    L.tc_attn_fwd_seg(q.data_ptr(), d, kv.data_ptr(), 2 * d, kv[:, d:].data_ptr(), 2 * d, Nk * 2 * d, o.data_ptr(), d, lse.data_ptr(), B, len(nq), nqc, Nk, 0.125, 0, TC_BF16, st)
    torch.cuda.synchronize()
    return o.float(), lse
torch.manual_seed(0)
B, nq, Nk = 1, [64], 64
q = torch.randn(64, d, device=dev).bfloat16(); k = torch.randn(Nk, d, device=dev).bfloat16(); v = torch.randn(Nk, d, device=dev).bfloat16()
o, lse = run(q, k, torch.ones_like(v), B, nq, Nk); print("V=1: O min/max", o.min().item(), o.max().item())
o, lse = run(q, torch.zeros_like(k), v, B, nq, Nk); ref = v.float().mean(0); print("K=0: max|O-mean V|", (o - ref).abs().max().item(), "lse", lse[:4].tolist(), "expect", torch.log(torch.tensor(64.0)).item())
# one-hot V rows: O[q][c] = P[q][key c] for keys c < 64
eye = torch.eye(64, device=dev).bfloat16()
o, lse = run(q, k, eye, B, nq, Nk)
s = (q.float() @ k.float().T) * 0.125; p = torch.softmax(s, -1)
err = (o - p).abs()
print("V=I: max|O-P|", err.max().item(), "per key-column max err:", [round(x, 3) for x in err.max(0).values.tolist()])
print("per query-row max err:", [round(x, 3) for x in err.max(1).values.tolist()])
print("lse err", (lse - torch.logsumexp(s, -1)).abs().max().item())
