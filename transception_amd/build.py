"""Builds libtransception_hip.so (gfx950) in-tree with hipcc.  No GPU needed: hipcc cross-compiles.

    python -m transception_amd.build [--force]
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtransception_hip.so")
SOURCES = ["gemm.hip", "norm.hip", "dwconv.hip", "softmax.hip", "attention.hip", "attention_seg.hip", "elementwise.hip", "train.hip", "factoratt.hip", "data.hip", "mixffn.hip", "mixffn_bwd.hip", "effatt.hip", "cbam.hip", "ripm.hip", "linln.hip", "lncls.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-munsafe-fp-atomics"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    # the hand-scheduled attention stream is generated source: refresh attn_fwd_asm.inc (rewritten only when its text changes)
    # (the generators read TC_ATTN_* knobs -- ablation modes that change results among them -- for the experiment builds under scripts/exp;
    # a product build must not pick them up from the caller's environment and overwrite the committed streams: ADVICE r4)
    genv = {k: v for k, v in os.environ.items() if not k.startswith(("TC_ATTN_", "TC_AS_", "TC_DKV_", "TC_DQ_"))}
    for gen in ("gen_attn_asm.py", "gen_dkv_asm.py", "gen_dq_asm.py"):
        r = subprocess.run([sys.executable, os.path.join(CSRC, gen)], capture_output=True, text=True, env=genv)
        if r.returncode != 0:
            raise RuntimeError(f"{gen} failed:\n{r.stderr}")
    headers = [os.path.join(CSRC, "tc_common.h"), os.path.join(HERE, "..", "include", "transception_hip.h")]
    extra = {"attention_seg.hip": [os.path.join(CSRC, "attn_fwd_asm.inc"), os.path.join(CSRC, "attn_dkv_asm.inc"), os.path.join(CSRC, "attn_dq_asm.inc")]}
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers + extra.get(src, [])):
            jobs.append((s, o))

    # per-file flags.  lncls.hip: hipcc's SLP vectoriser packs the classifier's fp32 multiply-adds into v_pk_fma_f32 / v_pk_add_f32 and pays
    # for it in v_mov_b32 (305 of the forward kernel's 1412 instructions; a v_pk_fma_f32 takes 5.0 SIMD-cycles against 3.0 for a v_fma_f32, scripts/exp/valu_rate.hip): 8 % fewer instructions without
    extra_flags = {"lncls.hip": ["-fno-slp-vectorize"]}

    def compile_one(job):
        s, o = job
        cmd = [hipcc, *FLAGS, *extra_flags.get(os.path.basename(s), []), "-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stderr}")
        return s

    with cf.ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        for s in ex.map(compile_one, jobs):
            if verbose:
                print(f"[build] compiled {os.path.basename(s)}")
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print(f"[build] linked {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
