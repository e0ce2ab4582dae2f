"""Per-wave s_memtime stamps of the hand-scheduled attention forward (libattn_timing.so = csrc/attention_seg.hip built with
TC_ATTN_TIMING=1 python csrc/gen_attn_asm.py): entry, asm start, prologue end, first body end, loop end, drain end, epilogue end, exit."""
import ctypes as C, os, sys, numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
import glob
dev = torch.device("cuda:0")
B, nq, Nk, d = 16, [3136, 1568, 980, 392], 784, 64
rows = B * sum(nq)
q = torch.randn(rows, d, device=dev).bfloat16(); kv = torch.randn(B * Nk, 2 * d, device=dev).bfloat16()
o = torch.empty_like(q); lse = torch.empty(rows, device=dev)
nqc = (C.c_int * 4)(*nq); st = torch.cuda.current_stream().cuda_stream
k, v = kv[:, :d], kv[:, d:]
names = ["stage Q,K,V + barrier", "asm prologue (Q frags, zero acc, QK(0))", "first iteration", "loop (nsub-2 iterations)", "last + drain", "epilogue (O -> LDS)", "stores"]
for so in sorted(glob.glob(os.path.join(here, "libattn_timing_*.so"))):
  L = C.CDLL(so)
  f = L.tc_attn_fwd_seg
  f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]
  for _ in range(3):
    f(q.data_ptr(), d, k.data_ptr(), 2 * d, v.data_ptr(), 2 * d, Nk * 2 * d, o.data_ptr(), d, lse.data_ptr(), B, 4, nqc, Nk, 0.125, 1, 1, st)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(20): f(q.data_ptr(), d, k.data_ptr(), 2 * d, v.data_ptr(), 2 * d, Nk * 2 * d, o.data_ptr(), d, lse.data_ptr(), B, 4, nqc, Nk, 0.125, 1, 1, st)
  e1.record(); torch.cuda.synchronize()
  us = e0.elapsed_time(e1) * 1e3 / 20
  buf = np.zeros(256 * 12 * 8, dtype=np.uint64)
  L.tc_attn_dbg_read.argtypes = [C.c_void_p]
  L.tc_attn_dbg_read(buf.ctypes.data)
  t = buf.reshape(256 * 12, 8).astype(np.int64)
  t[:, 1:7] &= 0xffffffff                      # (neighbouring lanes overlap the high word: the low words are the stamps)
  t[:, 0] &= 0xffffffff; t[:, 7] &= 0xffffffff
  dt = np.diff(t, axis=1) & 0xffffffff
  print(os.path.basename(so), f"{us:.1f} us per launch (events, back to back)")
  for i, n in enumerate(names):
    print(f"  {n:45s} {np.median(dt[:, i]):9.0f} {dt[:, i].min():9d} {dt[:, i].max():9d}")
  tot = (t[:, 7] - t[:, 0]) & 0xffffffff
  if "--by-wave" in sys.argv:                 # loop cycles by wave index (mean over the 256 workgroups): which waves of a SIMD run ahead
    lw = dt[:, 3].reshape(256, 12)
    print("  loop cycles by wave (mean over workgroups):", " ".join(f"{x:.0f}" for x in lw.mean(axis=0)))
    print("  start of the loop relative to the workgroup's first wave, by wave:", " ".join(f"{x:.0f}" for x in ((t[:, 3].reshape(256, 12) - t[:, 3].reshape(256, 12).min(axis=1, keepdims=True)) & 0xffffffff).mean(axis=0)))
  print(f"  per loop iteration (one 32-key sub-tile of one wave): {np.median(dt[:, 3]) / 23:.0f} cycles; whole wave {np.median(tot):.0f}")
sys.exit(0)
f = L.tc_attn_fwd_seg
f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]
k, v = kv[:, :d], kv[:, d:]
for _ in range(3):
    f(q.data_ptr(), d, k.data_ptr(), 2 * d, v.data_ptr(), 2 * d, Nk * 2 * d, o.data_ptr(), d, lse.data_ptr(), B, 4, nqc, Nk, 0.125, 1, 1, st)
torch.cuda.synchronize()
buf = np.zeros(256 * 12 * 8, dtype=np.uint64)
L.tc_attn_dbg_read.argtypes = [C.c_void_p]
L.tc_attn_dbg_read(buf.ctypes.data)
t = buf.reshape(256 * 12, 8).astype(np.int64)
dt = np.diff(t, axis=1)
names = ["stage Q,K,V + barrier", "asm prologue (Q frags, zero acc, QK(0))", "first iteration", "loop (nsub-2 iterations)", "last + drain", "epilogue (O -> LDS)", "stores"]
print("cycles per phase: median / min / max over 1024 waves (s_memtime ticks; 100 MHz?? see total)")
for i, n in enumerate(names):
    print(f"  {n:45s} {np.median(dt[:, i]):9.0f} {dt[:, i].min():9d} {dt[:, i].max():9d}")
tot = t[:, 7] - t[:, 0]
print("  total per wave", np.median(tot), tot.min(), tot.max(), " kernel span", t[:, 7].max() - t[:, 0].min())
print("  per loop body:", np.median(dt[:, 3]) / 23)
