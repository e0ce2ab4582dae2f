"""Runs the bridge attention forward back to back for a few seconds and samples rocm-smi (sclk, power) meanwhile: is the stream clock- / power-limited?"""
import ctypes as C, subprocess, sys, threading, time, torch
sys.path.insert(0, "/root/repo")
from transception_amd._lib import lib, TC_BF16
dev = torch.device("cuda:0")
libs = [a for a in sys.argv[1:] if a.endswith(".so")]            # variant builds (scripts/exp/build_attn_timing.sh with TC_ATTN_TIMING=0); default: the product library
if libs:
    L = C.CDLL(libs[0])
    L.tc_attn_fwd_seg.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]
else:
    L = lib()
B, nq, Nk, d = 16, [3136, 1568, 980, 392], 784, 64
rows = B * sum(nq)
q = torch.randn(rows, d, device=dev).bfloat16(); kv = torch.randn(B * Nk, 2 * d, device=dev).bfloat16()
if "--zeros" in sys.argv: q.zero_(); kv.zero_()
o = torch.empty_like(q); lse = torch.empty(rows, device=dev); nqc = (C.c_int * 4)(*nq)
k, v = kv[:, :d], kv[:, d:]
def fwd(st=None):
    st = torch.cuda.current_stream().cuda_stream
    L.tc_attn_fwd_seg(q.data_ptr(), d, k.data_ptr(), 2 * d, v.data_ptr(), 2 * d, Nk * 2 * d, o.data_ptr(), d, lse.data_ptr(), B, 4, nqc, Nk, 0.125, 1, TC_BF16, st)
if "--bwd" in sys.argv:                         # the backward call instead (dQ stream + dK/dV stream + partial fold), product library only
    do = torch.randn(rows, d, device=dev).bfloat16(); dq = torch.empty_like(q); dkv = torch.empty_like(kv)
    if "--zeros" in sys.argv: do.zero_()
    fwd(); torch.cuda.synchronize()
    delta = torch.empty(rows, device=dev); dkv32 = torch.empty(8 * B * Nk * 128, device=dev)
    def fwd(st=None):
        st = torch.cuda.current_stream().cuda_stream
        L.tc_attn_bwd_seg(q.data_ptr(), d, k.data_ptr(), 2 * d, v.data_ptr(), 2 * d, Nk * 2 * d, o.data_ptr(), d, do.data_ptr(), d, lse.data_ptr(), delta.data_ptr(), dkv32.data_ptr(),
                          dq.data_ptr(), d, dkv.data_ptr(), 2 * d, dkv[:, d:].data_ptr(), 2 * d, Nk * 2 * d, B, 4, nqc, Nk, 0.125, 1, TC_BF16, st)
g = torch.cuda.CUDAGraph()
for _ in range(3): fwd()
torch.cuda.synchronize()
with torch.cuda.graph(g):
    for _ in range(50 if '--bwd' in sys.argv else 200): fwd()
out = []
def sample():
    for _ in range(2):
        time.sleep(1.2)
        r = subprocess.run("rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E 'sclk|Power|Temp' | head -8", shell=True, capture_output=True, text=True)
        out.append(r.stdout)
th = threading.Thread(target=sample); th.start()
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < 3.5:
    g.replay(); n += (50 if '--bwd' in sys.argv else 200)
    if n % 4000 == 0: torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize(); th.join()
import re
clk = [int(m) for s_ in out for m in re.findall(r"\((\d+)Mhz\)", s_)]; pw = [float(m) for s_ in out for m in re.findall(r"Power \(W\): ([\d.]+)", s_)]
print(f"{(libs or ['product'])[0].split('/')[-1]:44s} {e0.elapsed_time(e1) * 1e3 / n:6.2f} us per launch sustained ({n} launches); sclk {clk} MHz, power {pw} W{' [zeros]' if '--zeros' in sys.argv else ''}{' [backward call]' if '--bwd' in sys.argv else ''}")
