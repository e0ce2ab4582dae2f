"""Per-phase cycle sums of the MixFFN mid-backward kernel (library built with -DTC_MID_TIMING): wave 0 of the first
workgroups of chunk 0.  usage: TC_LIB_PATH=scripts/exp/libtc_midtiming.so python scripts/exp/mid_timing.py C B H W groups"""
import ctypes as C, os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
args = sys.argv[1:6] or ["64", "16", "56", "56", "1"]
sys.argv = [sys.argv[0]] + args + ["--reps", "2", "--only", "fused"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench_ffn.py")).read().replace('if __name__ == "__main__":', "if True:"))
from transception_amd._lib import lib
L = lib()._dll if hasattr(lib(), "_dll") else C.CDLL(os.environ["TC_LIB_PATH"])
buf = np.zeros(64 * 16, dtype=np.int64)
f = C.CDLL(os.environ["TC_LIB_PATH"]).tc_mid_dbg_read
f.argtypes = [C.c_void_p]
print("rc", f(buf.ctypes.data))
t = buf.reshape(64, 16)
names = ["0 prologue (taps, first fetch issued, sync)", "1 top barrier (prev readers done)", "2 row statistics -> LDS, sync", "3 LN-backward of the tile into LDS (waits on the prefetched loads), sync",
         "4 issue next tile's loads", "5 dh = conv^T(dd) + dd, stores", "10 weight-gradient products", "6 (after loop)", "7 final barrier", "8 LDS atomics + sync", "9 fold + global atomics"]
idx = [0, 1, 2, 3, 4, 5, 10, 6, 7, 8, 9]
tot = t[:8, :].sum(1)
print("workgroups 0..7 total cycles:", tot)
for n, i in zip(names, idx):
    print(f"  {n:90s} {t[:8, i].mean():10.0f}  ({100.0 * t[:8, i].mean() / tot.mean():5.1f} %)")
