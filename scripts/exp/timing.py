import ctypes as C, os, torch, numpy as np
dev = torch.device("cuda:0")
B, nq, Nk, d = 16, [3136, 1568, 980, 392], 784, 64
rows = B * sum(nq)
torch.manual_seed(0)
q = torch.randn(rows, d, device=dev).bfloat16(); kv = torch.randn(B * Nk, 2 * d, device=dev).bfloat16()
lse = torch.empty(rows, device=dev); o = torch.zeros_like(q)
nqc = (C.c_int * 4)(*nq); st = torch.cuda.current_stream().cuda_stream
k, v = kv[:, :d], kv[:, d:]
L = C.CDLL(os.path.dirname(os.path.abspath(__file__)) + "/libatt_timing.so"); f = L.tc_attn_fwd_seg
f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
for _ in range(5):
    f(q.data_ptr(), d, k.data_ptr(), 2 * d, v.data_ptr(), 2 * d, Nk * 2 * d, o.data_ptr(), d, lse.data_ptr(), B, 4, nqc, Nk, 0.125, 1, st)
torch.cuda.synchronize()
NW = int(os.environ.get("NW", "12"))
buf = np.zeros(4 * NW * 24, dtype=np.int64)
L.tc_dbg_read.argtypes = [C.c_void_p]
print("rc", L.tc_dbg_read(buf.ctypes.data))
t = buf.reshape(4, NW, 24)
for blk in range(2):
    t0 = t[blk, :, 18].min() if t[blk, :, 18].min() > 0 else t[blk, :, 0].min()
    print(f"block {blk}: per wave (simd = wave%4): start, after-first-barrier, then per tile [arrive, leave], end   (cycles from block start)")
    for w in range(NW):
        r = t[blk, w] - t0
        tiles = " ".join(f"[{r[2+2*i]:6d} +{r[3+2*i]-r[2+2*i]:5d}]" for i in range(7))
        print(f"  w{w:2d} launch {r[18]:5d} q-issued {r[0]:5d} bar0 {r[1]:6d} {tiles} loop-end {r[19]:6d} end {r[20]:6d}")
