"""Per-phase cycle sums of the tiled MixFFN backward kernels (library built with -DTC_FFNB_TIMING: scripts/build_variant.sh
libtc_ffnb.so mixffn_bwd.hip -DTC_FFNB_TIMING): thread 0 of the first 16 workgroups of weight group 0, last launch of each kernel.
usage: TC_LIB_PATH=transception_amd/libtc_ffnb.so python scripts/exp/ffnb_timing.py C B H W groups"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
args = sys.argv[1:6] or ["64", "16", "56", "56", "1"]
sys.argv = [sys.argv[0]] + args + ["--reps", "2", "--only", "fused"]
exec(open(os.path.join(ROOT, "scripts", "bench_ffn.py")).read().replace('if __name__ == "__main__":', "if True:"))
buf = np.zeros(2 * 16 * 16, dtype=np.int64)
try:                                                   # (a forward-only timing build has no backward table)
    f = C.CDLL(os.environ["TC_LIB_PATH"]).tc_ffnb_dbg_read
    f.argtypes = [C.c_void_p]
    print("rc", f(buf.ctypes.data))
except AttributeError:
    pass
t = buf.reshape(2, 16, 16)
N1 = ["0 prologue (W2 fragments, gamma/beta, first fetch)", "1 put (waits on the fetch)", "2 barrier", "3 gpre MFMA", "4 epilogue (LN, GELU, GELU', sums, a -> LDS)", "5 barrier",
      "6 next fetch issue + dW2 MFMA", "7 LayerNorm backward -> LDS", "8 barrier", "9 gd store", "10 barrier", "11 tail (partials)"]
N2 = ["0 prologue (taps, W1 -> LDS, first fetch)", "1 xput/gput (waits on the fetch)", "2 barrier", "3 next fetch issue + fc1 MFMA -> h", "4 barrier", "5 dw stage (dwd sums, dh in registers)",
      "6 barrier", "7 dh -> LDS, LDS atomics", "8 barrier", "9 dx / dW1 MFMA", "10 barrier", "11 dx stage + store + barrier", "12 tail (partials)"]
try:
    ff = C.CDLL(os.environ["TC_LIB_PATH"]).tc_ffnf_dbg_read
    ff.argtypes = [C.c_void_p]
    fb = np.zeros(16 * 16, dtype=np.int64)
    ff(fb.ctypes.data)
    tf = fb.reshape(16, 16)
    NF = ["0 prologue (parameters, W1 fragments, first x tile)", "1 x prefetch issue + fc1 MFMA -> h", "2 dw3x3 + skip + LayerNorm partials", "3 barrier", "4 row statistics + barrier",
          "5 GELU(LN(d)) in place, d / statistics leave", "6 barrier", "7 fc2 MFMA (W2 fragments from L2)", "8 barrier", "9 out stage + barrier", "10 + bias + residual, out store", "11 xput + barrier"]
    tot = tf.sum(1)
    print(f"forward: workgroups 0..15 total cycles: mean {tot.mean():.0f} min {tot.min()} max {tot.max()}")
    for i, n in enumerate(NF):
        print(f"  {n:70s} {tf[:, i].mean():10.0f}  ({100.0 * tf[:, i].mean() / max(tot.mean(), 1):5.1f} %)")
except AttributeError:
    pass
for k, names in ((0, N1), (1, N2)):
    tot = t[k].sum(1)
    print(f"kernel {k + 1}: workgroups 0..15 total cycles: mean {tot.mean():.0f} min {tot.min()} max {tot.max()}")
    for i, n in enumerate(names):
        print(f"  {n:70s} {t[k, :, i].mean():10.0f}  ({100.0 * t[k, :, i].mean() / max(tot.mean(), 1):5.1f} %)")
