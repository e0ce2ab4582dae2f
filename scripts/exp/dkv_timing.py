"""Phase cycles of attn_bwd_dkv_seg_kernel (library built with -DTC_DKV_TIMING) at the bench shape.
usage: TC_LIB_PATH=scripts/exp/libtc_dkvtiming.so python scripts/exp/dkv_timing.py"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
exec(open(os.path.join(ROOT, "scripts", "bench_attn.py")).read())
L2 = C.CDLL(os.environ["TC_LIB_PATH"]); L2.tc_dkv_dbg_read.argtypes = [C.c_void_p]
buf = np.zeros(512 * 4, dtype=np.uint64)
L2.tc_dkv_dbg_read(buf.ctypes.data)
t = buf.reshape(512, 4).astype(np.float64)
t = t[t[:, 1] > 0]
print(len(t), "workgroups: prologue (K/V fragments, first stage) %.0f  query loop %.0f  epilogue (transpose + atomics) %.0f cycles" % (t[:, 0].mean(), t[:, 1].mean(), t[:, 2].mean()))
