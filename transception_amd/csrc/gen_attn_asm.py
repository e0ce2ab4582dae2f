#!/usr/bin/env python3
"""Emits the hand-scheduled gfx950 instruction stream of the bridge SR-attention forward (attn_fwd_asm_kernel in attention_seg.hip).

Why a generator.  The compiler-scheduled kernel (attn_fwd_seg_kernel) runs its twelve waves per CU in lock step -- QK^T MFMAs,
then softmax VALU, then PV MFMAs, re-aligned by a barrier every 128 keys -- so the matrix pipe idles while the VALU works and
vice versa (0.265 of the bf16 MFMA peak in rounds 2-3, and hipcc would not keep a software-pipelined order: variants v60/v74/v80/v92 of
rounds 2-3, `git show 970a2e3:scripts/exp/attn_exp.hip`).  Measured on gfx950 this round (scripts/exp/overlap2.hip, profiles/r4_overlap2.txt): one wave issues
about one instruction per 5.4 cycles whatever its type, so a single wave per SIMD is issue-bound (the first version of this file:
3 query tiles per wave, 486 cycles per 8 MFMAs); but the VALU stream of one wave DOES run under the MFMAs of another wave of the same
SIMD (8 MFMAs 256 -> 269 cycles next to a VALU wave that loses 15 %).  Hence: twelve waves per CU, ONE 32-query tile per wave, and
every wave runs the same software-pipelined stream in which no phase is MFMA-only or VALU-only:

  iteration j:  PV(j)   4 MFMAs, carrying the softmax of sub-tile j+1 (exp2 in place, row sum, overflow check) and the K(j+2) fragment reads
                pack P(j+1)
                QK(j+2) 4 MFMAs, carrying the V(j+1) fragment reads, the ring store of sub-tile j+3 and the loop bookkeeping

Scores come out of the QK^T MFMAs as s * scale * log2(e) - m (Q is scaled when it is staged, -m is the C operand of the first
MFMA), so the softmax is exp2 + add + pack per score and one compare per sub-tile.  The stream is spelled as a small macro assembler with two checks a human
cannot be trusted with:
  * counted s_waitcnt lgkmcnt()/vmcnt() come from a model of the in-order return queues, and the loop back-edge is verified to
    reproduce the queue state the loop body was generated for;
  * the MFMA / transcendental / permlane wait-state rules hipcc's hazard recogniser applies (inline asm gets none of them) are
    enforced with s_nop along the static order and re-checked on the dynamic traces.

Reference semantics: softmax(Q K^T * scale) V of `M_EfficientSelfAtten` (networks/MSTr.py:2281-2287).  Numerics: a fixed integer
reference exponent m per query row, set from the first key sub-tile and raised whenever a score exceeds it by more than REREF
(a 16-key partial row sum reaching 2^30 for bf16 P, 2^14 for fp16 P, or inf / NaN) -- the scheme of attn_fwd_seg_kernel; the
rare path recomputes the score tile (it was exponentiated in place) against the raised reference.

    python gen_attn_asm.py            # writes attn_fwd_asm.inc next to this file
"""
from __future__ import annotations

import os
import sys

# ---------------------------------------------------------------------------------------------------------------------------------
# layout shared with attn_fwd_asm_kernel (keep in sync with the AS_* constants there)
LDR_B = 144                     # bytes per LDS row: 64 halfs + 8 pad
SLOT_B = 2 * 32 * LDR_B          # ring slot: K sub-tile | V sub-tile of 32 keys
NSLOT = 12                      # slots; sub-tile j+AHEAD is stored during iteration j, and the workgroup barrier comes every PERIOD iterations:
AHEAD = int(os.environ.get("TC_ATTN_AHEAD", "6"))   #   a store must be separated from the first read of its slot (2 iterations before its PV) and from the last
PERIOD = 4                      #   read of the slot's previous occupant by a barrier: AHEAD >= PERIOD + 2, NSLOT >= AHEAD + PERIOD - 1 (+ slack)
NEG_BIG = 0xF149F2CA            # -1.0e30f
DEFER = bool(int(os.environ.get("TC_ATTN_DEFER", "0")))   # experiment: ring store of the sub-tile requested ONE iteration ago, at the head of the
#   iteration (stores run AHEAD - 1 ahead; with TC_ATTN_AHEAD=7 and -DTC_AS_AHEAD=7, scripts/exp/build_fwd_variant.sh): 25.6 vs 25.7 us -- the
#   wait for the global load is not what an iteration waits for

TIMING = bool(int(os.environ.get("TC_ATTN_TIMING", "0")))       # experiment builds: s_memtime stamps at section boundaries
NSTAMP = 6
ABLATE = os.environ.get("TC_ATTN_ABLATE", "")                    # timing experiments only (wrong results): noexp, novalu, nomfma, nostage, nobar


def vreg(n, w=1):
    return f"v{n}" if w == 1 else f"v[{n}:{n + w - 1}]"


def areg(n, w=1):
    return f"a{n}" if w == 1 else f"a[{n}:{n + w - 1}]"


def regs(prefix, n, w=1):
    return [f"{prefix}{i}" for i in range(n, n + w)]


# register map (the VGPRs / AGPRs the stream owns; the kernel's own values live above NV) -------------------------------------------
S = 0                                        # score tile s * qs - m, then P in fp32 (exp2 in place)          16
NEGM = 16                                    # -m broadcast: the C operand of the first QK^T MFMA            16
P = 32                                       # packed P: two B operands of 4                                  8
ST = 40                                      # ring staging (one 16-byte chunk per thread of waves 0-7)       4
L, M, RS, T0, T1, T2, T3, T4 = 44, 45, 46, 47, 48, 49, 50, 51
AK, AV, AW, MSK = 52, 53, 54, 55             # LDS addresses of the current ring slots (the subroutines own T0..T4); tail mask
NV = 56
# AGPRs (at three waves per SIMD hipcc splits the 168 registers of a wave 84 / 84, so the MFMA-only operands live here)
def O(blk): return 16 * blk                  # O^T accumulators                                              32
def QF(ks): return 32 + 4 * ks               # Q fragments                                                   16
def KF(ks): return 48 + 4 * ks               # K fragments of the next sub-tile                              16
def VF(f): return 64 + 4 * f                 # V^T fragments, f = 2 * k2 + blk                                16
NA = 80
# operands of the asm statement: %0 is the only output ("+v": the global offset of this thread's staging chunk, advanced per iteration)
OP_GOFF, OP_KBASE, OP_VBASE, OP_WBASE, OP_QADDR, OP_OADDR, OP_LSEADDR, OP_MASK, OP_RSRC, OP_NSUB, OP_STEP, OP_WEXEC = (f"%{i}" for i in range(12))
S_CNT, S_THR, S_SV, S_SK, S_SW, S_PH, S_RA, S_LN2, S_STEP = 60, 61, 62, 63, 64, 65, 66, 68, 69
SGPRS = list(range(60, 70))
if TIMING:
    SGPRS = list(range(60, 72 + 2 * NSTAMP))


class Ins:
    __slots__ = ("text", "kind", "rd", "wr", "ws")

    def __init__(self, text, kind, rd=(), wr=(), ws=1):
        self.text, self.kind, self.rd, self.wr, self.ws = text, kind, tuple(rd), tuple(wr), ws


class Gen:
    """Instruction list + in-order memory-return queues.  Everything is appended through emit(); waits are inserted on demand."""

    def __init__(self, half):
        self.half = half                    # 'bf16' | 'f16'
        self.out = []                       # Ins, static program order
        self.lgkm = []                      # outstanding LDS ops: tuple of written regs (empty for stores)
        self.vm = []
        self.uid = 0
        self.t = 0                          # wait-state clock of the static order, for the hazard padding below
        self.last_mfma, self.last_valu, self.last_trans = {}, {}, {}

    # -- queues ---------------------------------------------------------------------------------------------------------------
    def _need(self, q, name, touched):
        idx = -1
        for i, wr in enumerate(q):
            if any(r in touched for r in wr):
                idx = i
        if idx >= 0:
            n = min(len(q) - idx - 1, 15 if name == "lgkmcnt" else 63)     # the counters are 4 / 6 bits wide: a smaller count waits for more
            self._push(Ins(f"s_waitcnt {name}({n})", "wait"))
            del q[: len(q) - n]

    def wait_lgkm(self, n):
        n = min(n, 15)
        if len(self.lgkm) > n:
            self._push(Ins(f"s_waitcnt lgkmcnt({n})", "wait"))
            del self.lgkm[: len(self.lgkm) - n]

    def wait_vm(self, n):
        if len(self.vm) > n:
            self._push(Ins(f"s_waitcnt vmcnt({n})", "wait"))
            del self.vm[: len(self.vm) - n]

    def _push(self, ins):
        self.out.append(ins)
        self.t += ins.ws

    def _pad(self, kind, rd, wr):
        """s_nop padding for the wait-state rules of check_hazards(), along the static order (what hipcc's hazard recogniser does)."""
        need = 0
        def gap(tbl, r, n):
            return n - (self.t - tbl[r]) + 1 if r in tbl else 0
        if kind in ("valu", "trans", "permlane", "ds_read", "ds_write", "vmem_load"):
            for r in list(rd) + list(wr):
                need = max(need, gap(self.last_mfma, r, 11))
        if kind == "mfma":
            for r in rd:
                need = max(need, gap(self.last_valu, r, 2))
                if r not in wr:
                    need = max(need, gap(self.last_mfma, r, 11))
        if kind in ("valu", "permlane"):
            for r in rd:
                need = max(need, gap(self.last_trans, r, 1))
        if kind == "permlane":
            for r in rd:
                need = max(need, gap(self.last_valu, r, 2))
        self.nop(need)

    def _mark(self, kind, wr):
        for r in wr:
            if kind == "mfma":
                self.last_mfma[r] = self.t
                self.last_valu.pop(r, None)
            else:
                self.last_mfma.pop(r, None)
                if kind in ("valu", "trans", "permlane"):
                    self.last_valu[r] = self.t
            if kind == "trans":
                self.last_trans[r] = self.t
            else:
                self.last_trans.pop(r, None)

    def emit(self, text, kind, rd=(), wr=(), ws=1):
        touched = set(rd) | set(wr)
        self._need(self.lgkm, "lgkmcnt", touched)
        self._need(self.vm, "vmcnt", touched)
        self._pad(kind, rd, wr)
        self._mark(kind, wr)
        self._push(Ins(text, kind, rd, wr, ws))
        if kind == "ds_read":
            self.lgkm.append(tuple(wr))
        elif kind == "ds_write":
            self.lgkm.append(())
        elif kind == "vmem_load":
            self.vm.append(tuple(wr))

    def label(self, name):
        self.out.append(Ins(f"{name}:", "label", ws=0))

    def state(self):
        return (tuple(self.lgkm), tuple(self.vm))

    # -- instruction helpers ----------------------------------------------------------------------------------------------------
    def mfma(self, d, a, b, c, acc_d=False, b_acc=False):
        dn = regs("a" if acc_d else "v", d, 16)
        an = regs("a", a, 4)
        bn = regs("a" if b_acc else "v", b, 4)
        cn = regs("a" if acc_d else "v", c, 16)
        if "nomfma" in ABLATE:
            return
        self.emit(f"v_mfma_f32_32x32x16_{self.half} {(areg if acc_d else vreg)(d, 16)}, {areg(a, 4)}, {(areg if b_acc else vreg)(b, 4)}, "
                  f"{(areg if acc_d else vreg)(c, 16)}", "mfma", an + bn + cn, dn)

    def valu(self, text, rd=(), wr=(), trans=False):
        if trans and "noexp" in ABLATE:
            text, trans = text.replace("v_exp_f32_e32", "v_mov_b32_e32"), False
        self.emit(text, "trans" if trans else "valu", rd, wr)

    def salu(self, text):
        self.emit(text, "salu")

    def stamp(self, k):
        if TIMING:
            self.wait_lgkm(0)
            self._push(Ins(f"s_memtime s[{72 + 2 * k}:{73 + 2 * k}]", "salu"))
            self._push(Ins("s_waitcnt lgkmcnt(0)", "wait"))

    def nop(self, n):
        while n > 0:
            k = min(n, 16)
            self._push(Ins(f"s_nop {k - 1}", "nop", ws=k))
            n -= k


# ---------------------------------------------------------------------------------------------------------------------------------
def mf_qk(g, ks):
    g.mfma(S, KF(ks), QF(ks), NEGM if ks == 0 else S, b_acc=True)


def mf_pv(g, f):
    k2, blk = f >> 1, f & 1
    g.mfma(O(blk), VF(f), P + 4 * k2, O(blk), acc_d=True)


def vf_load(g, f, addr):
    k2, blk = f >> 1, f & 1
    for e in range(2):
        g.emit(f"ds_read_b64_tr_b16 {areg(VF(f) + 2 * e, 2)}, {vreg(addr)} offset:{(2 * k2 + e) * LDR_B + 64 * blk}", "ds_read",
               [vreg(addr)], regs("a", VF(f) + 2 * e, 2))


def kf_load(g, ks, addr):
    a = addr if isinstance(addr, str) else vreg(addr)
    g.emit(f"ds_read_b128 {areg(KF(ks), 4)}, {a} offset:{32 * ks}", "ds_read", [a], regs("a", KF(ks), 4))


def cvt_name(half):
    return "v_cvt_pk_bf16_f32" if half == "bf16" else "v_cvt_pk_f16_f32"


def max_tree(g, dst):
    """dst = max over the 16 scores of the lane: 7 x v_max3 + 1 x v_max, as closures."""
    it = [lambda: g.valu(f"v_max3_f32 {vreg(dst)}, {vreg(S)}, {vreg(S + 1)}, {vreg(S + 2)}", regs("v", S, 3), [vreg(dst)])]
    for r in range(3, 15, 2):
        it.append(lambda r=r: g.valu(f"v_max3_f32 {vreg(dst)}, {vreg(dst)}, {vreg(S + r)}, {vreg(S + r + 1)}", [vreg(dst)] + regs("v", S + r, 2), [vreg(dst)]))
    it.append(lambda: g.valu(f"v_max_f32_e32 {vreg(dst)}, {vreg(dst)}, {vreg(S + 15)}", [vreg(dst), vreg(S + 15)], [vreg(dst)]))
    return it


def mask_items(g, src):
    """Keys past Nk (bits of the VGPR / operand `src`, one per score register) get the score -1e30."""
    it = [lambda: g.valu(f"v_mov_b32_e32 {vreg(T3)}, 0x{NEG_BIG:08x}", [], [vreg(T3)])]
    for r in range(16):
        it.append(lambda r=r: g.valu(f"v_bfe_i32 {vreg(T4)}, {src}, {r}, 1", [], [vreg(T4)]))
        it.append(lambda r=r: g.valu(f"v_bfi_b32 {vreg(S + r)}, {vreg(T4)}, {vreg(T3)}, {vreg(S + r)}", [vreg(T4), vreg(T3), vreg(S + r)], [vreg(S + r)]))
    return it


def exp_sum_items(g):
    """exp2 in place and the 16-key row sum into RS: 16 + 15 instructions, each sum two instructions behind its exp."""
    def ex(r): return lambda: g.valu(f"v_exp_f32_e32 {vreg(S + r)}, {vreg(S + r)}", [vreg(S + r)], [vreg(S + r)], trans=True)
    def ad(r): return lambda: g.valu(f"v_add_f32_e32 {vreg(RS)}, {vreg(RS)}, {vreg(S + r)}", [vreg(RS), vreg(S + r)], [vreg(RS)])
    it = [ex(0), ex(1), ex(2), lambda: g.valu(f"v_add_f32_e32 {vreg(RS)}, {vreg(S)}, {vreg(S + 1)}", regs("v", S, 2), [vreg(RS)])]
    for r in range(3, 16):
        it += [ex(r), ad(r - 1)]
    it.append(ad(15))
    return it


def sm_items(g, mode):
    """Softmax of the score tile in S as closures (one instruction each unless noted): keys past Nk masked ('last'); the reference
    exponent set from this tile ('first': unconditional call); exp2 in place and the row sum; then ('loop', 'last') ONE compare of
    the row sum against REREF -- a sum that large (or inf / NaN) means some score outgrew the reference exponent, and the rare
    path recomputes the tile against a raised one; L += RS."""
    it = []
    if "novalu" in ABLATE and mode == "loop":
        return it
    if mode == "last":
        it += mask_items(g, OP_MASK)
    if mode == "first":
        it.append(lambda: g.emit(f"s_call_b64 s[{S_RA}:{S_RA + 1}], .Lfirst_%=", "call"))
    it += exp_sum_items(g)
    if mode != "first":
        def check():                           # one unit: nothing may be scheduled between the branch and its target
            g.valu(f"v_cmp_ngt_f32_e32 vcc, s{S_THR}, {vreg(RS)}", [vreg(RS)], ["vcc"])
            g.emit(f"s_cbranch_vccz .Lskip{g.uid}_%=", "branch", ["vcc"])
            g.emit(f"s_call_b64 s[{S_RA}:{S_RA + 1}], .Lslow_%=", "call")
            g.label(f".Lskip{g.uid}_%=")
            g.uid += 1
        it.append(check)
    it.append(lambda: g.valu(f"v_add_f32_e32 {vreg(L)}, {vreg(L)}, {vreg(RS)}", [vreg(L), vreg(RS)], [vreg(L)]))
    return it


def pack_items(g):
    if "dot2" in ABLATE:
        d2 = "v_dot2c_f32_bf16" if "dot2c" in ABLATE else "v_dot2_f32_bf16"
        tail = "" if "dot2c" in ABLATE else f", {vreg(L)}"
        return [lambda q=q: g.valu(f"{cvt_name(g.half)} {vreg(P + q)}, {vreg(S + 2 * q)}, {vreg(S + 2 * q + 1)}", regs("v", S + 2 * q, 2), [vreg(P + q)])
                for q in range(8)] + [lambda q=q: g.valu(f"{d2} {vreg(L)}, {vreg(P + q)}, {vreg(T4)}{tail}", [vreg(P + q), vreg(T4), vreg(L)], [vreg(L)]) for q in range(8)]
    return [lambda q=q: g.valu(f"{cvt_name(g.half)} {vreg(P + q)}, {vreg(S + 2 * q)}, {vreg(S + 2 * q + 1)}", regs("v", S + 2 * q, 2), [vreg(P + q)])
            for q in range(8)]


def group(g, mfmas, free, head=()):
    """head: closures before the first MFMA; mfmas: closures; free: ordered closures spread evenly over the gaps after the MFMAs."""
    for f in head:
        f()
    n = len(mfmas)
    if n == 0:
        for f in free:
            f()
        return
    per = [len(free) // n + (1 if k < len(free) % n else 0) for k in range(n)]
    pos = 0
    for k, m in enumerate(mfmas):
        m()
        for f in free[pos: pos + per[k]]:
            f()
        pos += per[k]


def iteration(g, mode):
    """Iteration j: PV(j) | softmax(j+1) | K(j+2) reads; pack P(j+1); QK(j+2) | V(j+1) reads | ring store of sub-tile j+3.
    mode 'first' (j = -1): no PV, the softmax sets the reference exponent; 'loop'; 'last' (j = nsub-2): keys past Nk masked,
    no sub-tile j+2; 'drain' (j = nsub-1): PV only."""
    first, last, drain = mode == "first", mode == "last", mode == "drain"
    stage = not (last or drain) and not ("nostage" in ABLATE and mode == "loop")
    # ---- PV(j) group
    head = []
    if mode == "loop":                         # every PERIOD-th iteration: the stores of the last PERIOD iterations become visible, their
        def bar():                             # slots' previous occupants are dead
            g.wait_lgkm(0)
            if "nobar" not in ABLATE:
                g.salu(f"s_cmp_lg_u32 s{S_PH}, 0")
                g.emit(f"s_cbranch_scc1 .Lnobar{g.uid}_%=", "branch")
                g.emit("s_barrier", "barrier")
                g.label(f".Lnobar{g.uid}_%=")
                g.uid += 1
            g.salu(f"s_add_u32 s{S_PH}, s{S_PH}, 1")
            g.salu(f"s_and_b32 s{S_PH}, s{S_PH}, {PERIOD - 1}")
        head.append(bar)
    def stash():
        g.wait_vm(0)
        g.valu(f"v_add_u32_e32 {vreg(AW)}, s{S_SW}, {OP_WBASE}", [], [vreg(AW)])
        g.salu(f"s_mov_b64 exec, {OP_WEXEC}")
        g.emit(f"ds_write_b128 {vreg(AW)}, {vreg(ST, 4)}", "ds_write", [vreg(AW)] + regs("v", ST, 4))
        g.salu("s_mov_b64 exec, -1")
    if stage:
        if DEFER and not first:
            head.append(stash)
        head.append(lambda: g.salu(f"s_mov_b64 exec, {OP_WEXEC}"))
        head.append(lambda: g.emit(f"buffer_load_dwordx4 {vreg(ST, 4)}, {OP_GOFF}, {OP_RSRC}, 0 offen", "vmem_load", [], regs("v", ST, 4)))
        head.append(lambda: g.salu("s_mov_b64 exec, -1"))
        head.append(lambda: g.valu(f"v_add_u32_e32 {OP_GOFF}, s{S_STEP}, {OP_GOFF}", [], []))
    if last:
        head.append(lambda: g.valu(f"v_mov_b32_e32 {vreg(MSK)}, {OP_MASK}", [], [vreg(MSK)]))
    if not (last or drain):
        head.append(lambda: g.valu(f"v_add_u32_e32 {vreg(AK)}, s{S_SK}, {OP_KBASE}", [], [vreg(AK)]))
    mf = [] if first else [lambda f=f: mf_pv(g, f) for f in range(4)]
    free = [] if drain else sm_items(g, mode)
    if not (last or drain):                    # K(j+2) fragments: free since QK(j+1) issued in the previous iteration
        kl = [lambda ks=ks: kf_load(g, ks, AK) for ks in range(4)]
        free = free[:4] + kl + free[4:] if len(free) > 8 else kl + free
    if mf and "noprio" not in ABLATE:          # the group that carries this wave's VALU work runs at raised priority (measured -3 % loop time)
        head.append(lambda: g.salu("s_setprio 1"))
    group(g, mf, free, head)
    if mf and "noprio" not in ABLATE:
        g.salu("s_setprio 0")
    if drain:
        return
    for f in pack_items(g):
        f()
    # ---- QK(j+2) group
    head = [lambda: g.valu(f"v_add_u32_e32 {vreg(AV)}, s{S_SV}, {OP_VBASE}", [], [vreg(AV)])]
    mf = [] if last else [lambda ks=ks: mf_qk(g, ks) for ks in range(4)]
    free = [lambda f=f: vf_load(g, f, AV) for f in range(4)]      # V(j+1) fragments: free since PV(j) issued above
    if stage and not DEFER:
        free.append(stash)
    if not last:
        def slots():
            for s in (S_SV, S_SK, S_SW):      # ring slot offsets of the next iteration
                g.salu(f"s_add_u32 s{s}, s{s}, {SLOT_B}")
                g.salu(f"s_cmp_eq_u32 s{s}, {SLOT_B * NSLOT}")
                g.salu(f"s_cselect_b32 s{s}, 0, s{s}")
        free.append(slots)
    if "prioQ" in ABLATE and mf:
        head.append(lambda: g.salu("s_setprio 1"))
    group(g, mf, free, head)
    if "prioQ" in ABLATE and mf:
        g.salu("s_setprio 0")


def rereference(g, first):
    """S holds s * qs - m_old: m_delta = ceil(row max over both lane halves) (first) / max(that, 0); rescale L and O by 2^-m_delta
    (not first), m += m_delta, NEGM -= m_delta, S -= m_delta."""
    for f in max_tree(g, T0):
        f()
    g.valu(f"v_mov_b32_e32 {vreg(T1)}, {vreg(T0)}", [vreg(T0)], [vreg(T1)])
    g.emit(f"v_permlane32_swap_b32_e32 {vreg(T0)}, {vreg(T1)}", "permlane", [vreg(T0), vreg(T1)], [vreg(T0), vreg(T1)])
    g.valu(f"v_max_f32_e32 {vreg(T0)}, {vreg(T0)}, {vreg(T1)}", [vreg(T0), vreg(T1)], [vreg(T0)])
    g.valu(f"v_ceil_f32_e32 {vreg(T2)}, {vreg(T0)}", [vreg(T0)], [vreg(T2)])               # T2 = m_delta
    if not first:
        g.valu(f"v_max_f32_e32 {vreg(T2)}, 0, {vreg(T2)}", [vreg(T2)], [vreg(T2)])
        g.valu(f"v_sub_f32_e32 {vreg(T1)}, 0, {vreg(T2)}", [vreg(T2)], [vreg(T1)])
        g.valu(f"v_exp_f32_e32 {vreg(T3)}, {vreg(T1)}", [vreg(T1)], [vreg(T3)], trans=True)  # T3 = 2^-m_delta
        g.valu(f"v_mul_f32_e32 {vreg(L)}, {vreg(L)}, {vreg(T3)}", [vreg(L), vreg(T3)], [vreg(L)])
        for a in range(32):
            g.valu(f"v_accvgpr_read_b32 {vreg(T1)}, {areg(a)}", [areg(a)], [vreg(T1)])
            g.valu(f"v_mul_f32_e32 {vreg(T1)}, {vreg(T1)}, {vreg(T3)}", [vreg(T1), vreg(T3)], [vreg(T1)])
            g.valu(f"v_accvgpr_write_b32 {areg(a)}, {vreg(T1)}", [vreg(T1)], [areg(a)])
    g.valu(f"v_add_f32_e32 {vreg(M)}, {vreg(M)}, {vreg(T2)}", [vreg(M), vreg(T2)], [vreg(M)])
    for r in range(16):
        g.valu(f"v_sub_f32_e32 {vreg(NEGM + r)}, {vreg(NEGM + r)}, {vreg(T2)}", [vreg(NEGM + r), vreg(T2)], [vreg(NEGM + r)])
    for r in range(16):
        g.valu(f"v_sub_f32_e32 {vreg(S + r)}, {vreg(S + r)}, {vreg(T2)}", [vreg(S + r), vreg(T2)], [vreg(S + r)])


def subroutine(g, first):
    """Called, never fallen into; returns through s[S_RA:S_RA+1].
    first: S = QK(0) is intact (called before the exp2 pass): set the reference exponent from it; the caller's exp2 pass follows.
    slow:  called after the exp2 pass found a row sum >= REREF (or inf / NaN): the scores are gone, so reload the K fragments of
           this sub-tile (ring slot S_SV: sub-tile j+1 is both the V the next PV reads and the K of these scores), redo the four
           QK^T MFMAs against the old reference, mask the tail keys (MSK), raise the reference, redo exp2 / row sum, and put the
           K(j+2) fragments back (the caller's outstanding-load queue is exactly those four reads)."""
    g.label(f".L{'first' if first else 'slow'}_%=")
    g0 = Gen(g.half)                           # own wait-state / queue bookkeeping: entered and left with nothing assumed
    g0.uid = 1000
    g0.nop(32)                                 # every MFMA issued before the call has retired (S, O quiescent)
    if not first:
        g0._push(Ins("s_waitcnt lgkmcnt(0)", "wait"))
        g0.valu(f"v_add_u32_e32 {vreg(T0)}, s{S_SV}, {OP_KBASE}", [], [vreg(T0)])
        for ks in range(4):
            kf_load(g0, ks, T0)
        for ks in range(4):
            mf_qk(g0, ks)
        for f in mask_items(g0, vreg(MSK)):
            f()
    rereference(g0, first)
    if not first:
        for f in exp_sum_items(g0):
            f()
        g0.valu(f"v_add_u32_e32 {vreg(T0)}, s{S_SK}, {OP_KBASE}", [], [vreg(T0)])
        for ks in range(4):
            kf_load(g0, ks, T0)
    g0.nop(4)
    g0.emit(f"s_setpc_b64 s[{S_RA}:{S_RA + 1}]", "ret")
    g.out += g0.out


def prologue(g):
    g.salu(f"s_mov_b32 s{S_CNT}, {OP_NSUB}")
    g.salu(f"s_mov_b32 s{S_STEP}, {OP_STEP}")
    g.salu(f"s_mov_b32 s{S_THR}, {'0x4e800000' if g.half == 'bf16' else '0x46800000'}")    # REREF: a 16-key row sum of 2^30 (bf16 P) / 2^14 (fp16 P)
    g.salu(f"s_mov_b32 s{S_LN2}, 0x3f317218")
    g.salu(f"s_mov_b32 s{S_SV}, 0")                                                       # V(0)
    g.salu(f"s_mov_b32 s{S_SK}, {SLOT_B}")                                                # K(1)
    g.salu(f"s_mov_b32 s{S_SW}, {(AHEAD - (2 if DEFER else 1)) * SLOT_B}")               # sub-tile AHEAD-1 is stored by iteration -1 (DEFER: by iteration 0)
    g.salu(f"s_mov_b32 s{S_PH}, {2 % PERIOD}")                                            # (j + 2) mod PERIOD of iteration 0
    g.salu(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 2")                                            # steady iterations j = 0 .. nsub - 3
    for ks in range(4):
        g.emit(f"ds_read_b128 {areg(QF(ks), 4)}, {OP_QADDR} offset:{32 * ks}", "ds_read", [], regs("a", QF(ks), 4))
    for ks in range(4):
        kf_load(g, ks, OP_KBASE)                                                          # K(0): slot 0
    g.valu(f"v_mov_b32_e32 {vreg(L)}, 0", [], [vreg(L)])
    g.valu(f"v_mov_b32_e32 {vreg(M)}, 0", [], [vreg(M)])
    g.valu(f"v_mov_b32_e32 {vreg(MSK)}, 0", [], [vreg(MSK)])
    for r in range(16):
        g.valu(f"v_mov_b32_e32 {vreg(NEGM + r)}, 0", [], [vreg(NEGM + r)])
    for r in range(32):
        g.valu(f"v_accvgpr_write_b32 {areg(r)}, 0", [], [areg(r)])
    for ks in range(4):                                                                   # QK(0)
        mf_qk(g, ks)


def epilogue(g):
    g.nop(32)
    g.valu(f"v_mov_b32_e32 {vreg(T0)}, {vreg(L)}", [vreg(L)], [vreg(T0)])
    g.valu(f"v_mov_b32_e32 {vreg(T1)}, {vreg(L)}", [vreg(L)], [vreg(T1)])
    g.emit(f"v_permlane32_swap_b32_e32 {vreg(T0)}, {vreg(T1)}", "permlane", [vreg(T0), vreg(T1)], [vreg(T0), vreg(T1)])
    g.valu(f"v_add_f32_e32 {vreg(T0)}, {vreg(T0)}, {vreg(T1)}", [vreg(T0), vreg(T1)], [vreg(T0)])          # T0 = row sum over both halves
    g.valu(f"v_rcp_f32_e32 {vreg(T4)}, {vreg(T0)}", [vreg(T0)], [vreg(T4)], trans=True)
    g.valu(f"v_log_f32_e32 {vreg(T1)}, {vreg(T0)}", [vreg(T0)], [vreg(T1)], trans=True)
    g.valu(f"v_add_f32_e32 {vreg(T1)}, {vreg(T1)}, {vreg(M)}", [vreg(T1), vreg(M)], [vreg(T1)])
    g.valu(f"v_mul_f32_e32 {vreg(T1)}, s{S_LN2}, {vreg(T1)}", [vreg(T1)], [vreg(T1)])
    g.emit(f"ds_write_b32 {OP_LSEADDR}, {vreg(T1)}", "ds_write", [vreg(T1)])
    for blk in range(2):
        for q in range(4):
            for e in range(4):
                a = O(blk) + 4 * q + e
                g.valu(f"v_accvgpr_read_b32 {vreg(S + e)}, {areg(a)}", [areg(a)], [vreg(S + e)])
            for e in range(4):
                g.valu(f"v_mul_f32_e32 {vreg(S + e)}, {vreg(S + e)}, {vreg(T4)}", [vreg(S + e), vreg(T4)], [vreg(S + e)])
            g.valu(f"{cvt_name(g.half)} {vreg(S + 4)}, {vreg(S)}, {vreg(S + 1)}", regs("v", S, 2), [vreg(S + 4)])
            g.valu(f"{cvt_name(g.half)} {vreg(S + 5)}, {vreg(S + 2)}, {vreg(S + 3)}", regs("v", S + 2, 2), [vreg(S + 5)])
            g.emit(f"ds_write_b64 {OP_OADDR}, {vreg(S + 4, 2)} offset:{64 * blk + 16 * q}", "ds_write", [vreg(S + 4), vreg(S + 5)])
    g.wait_lgkm(0)


# ---------------------------------------------------------------------------------------------------------------------------------
def check_hazards(trace):
    """The wait-state rules of LLVM's GCNHazardRecognizer that matter here (gfx940/gfx950, 8-pass XDL MFMAs), on a dynamic trace:
      MFMA writes v/a  -> VALU / LDS / VMEM read or write of it : 11 wait states
      VALU writes v/a  -> MFMA reads it                          : 2
      trans writes v   -> non-trans VALU reads it                : 1
      VALU writes v    -> v_permlane reads it                    : 2
    (MFMA -> MFMA with C == D of the previous one issues back to back; A / B operands here never come from an MFMA.)"""
    last_mfma, last_valu, last_trans = {}, {}, {}
    t = 0
    for ins in trace:
        if ins.kind in ("label",):
            continue
        if ins.kind in ("valu", "trans", "permlane", "ds_read", "ds_write", "vmem_load"):
            for r in list(ins.rd) + list(ins.wr):
                if r in last_mfma and t - last_mfma[r] - 1 < 11:
                    raise RuntimeError(f"MFMA->{ins.kind} hazard on {r}: {ins.text} ({t - last_mfma[r] - 1} wait states)")
        if ins.kind == "mfma":
            for r in ins.rd:
                if r in last_valu and t - last_valu[r] - 1 < 2:
                    raise RuntimeError(f"VALU->MFMA hazard on {r}: {ins.text}")
                if r in last_mfma and r not in ins.wr and t - last_mfma[r] - 1 < 11:
                    raise RuntimeError(f"MFMA->MFMA A/B/C hazard on {r}: {ins.text}")
        if ins.kind in ("valu", "permlane"):
            for r in ins.rd:
                if r in last_trans and t - last_trans[r] - 1 < 1:
                    raise RuntimeError(f"trans->VALU hazard on {r}: {ins.text}")
        if ins.kind == "permlane":
            for r in ins.rd:
                if r in last_valu and t - last_valu[r] - 1 < 2:
                    raise RuntimeError(f"VALU->permlane hazard on {r}: {ins.text}")
        for r in ins.wr:
            if ins.kind == "mfma":
                last_mfma[r] = t
                last_valu.pop(r, None)
            else:
                last_mfma.pop(r, None)
                if ins.kind in ("valu", "trans", "permlane"):
                    last_valu[r] = t
            if ins.kind == "trans":
                last_trans[r] = t
            else:
                last_trans.pop(r, None)
        t += ins.ws


def generate(half):
    g = Gen(half)
    g.stamp(0)
    prologue(g)
    g.wait_lgkm(0)
    g.emit("s_barrier", "barrier")                       # the Q tiles (read by now) lie in ring slots that the stores of iteration 0.. reuse
    g.stamp(1)
    iteration(g, "first")
    g.stamp(2)
    first_end = len(g.out)
    st_in = g.state()
    g.salu(f"s_cmp_lt_i32 s{S_CNT}, 1")
    g.emit("s_cbranch_scc1 .Llast_%=", "branch")
    g.label(".Lloop_%=")
    loop_begin = len(g.out)
    iteration(g, "loop")
    if TIMING:
        g.wait_lgkm(0)                                   # (the stamp before the loop drained the queue)
    g.salu(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    g.salu(f"s_cmp_lg_u32 s{S_CNT}, 0")
    g.emit("s_cbranch_scc1 .Lloop_%=", "branch")
    loop_end = len(g.out)
    if g.state() != st_in:
        raise RuntimeError(f"loop back-edge changes the outstanding-load queues:\n in  {st_in}\n out {g.state()}")
    g.label(".Llast_%=")
    last_begin = len(g.out)
    g.stamp(3)
    g.nop(11)                                            # entered from the loop or straight from the first iteration (nsub = 2)
    iteration(g, "last")
    iteration(g, "drain")
    g.wait_lgkm(0)
    g.emit("s_barrier", "barrier")                       # the O tiles written below lie in ring slots other waves may still be reading
    g.stamp(4)
    epilogue(g)
    g.stamp(5)
    if TIMING:
        g.salu("s_mov_b64 exec, 1")                      # lane 0 only: the lanes' lse slots are 4 bytes apart
        for k in range(NSTAMP):
            g.valu(f"v_mov_b32_e32 {vreg(S)}, s{72 + 2 * k}", [], [vreg(S)])
            g.valu(f"v_mov_b32_e32 {vreg(S + 1)}, s{73 + 2 * k}", [], [vreg(S + 1)])
            g.emit(f"ds_write_b64 {OP_LSEADDR}, {vreg(S, 2)} offset:{4096 + 8 * k}", "ds_write", [vreg(S), vreg(S + 1)])
        g.wait_lgkm(0)
        g.salu("s_mov_b64 exec, -1")
    g.emit("s_branch .Lend_%=", "branch")
    main_end = len(g.out)
    subroutine(g, True)
    subroutine(g, False)
    g.label(".Lend_%=")
    main = g.out[:main_end]
    check_hazards(main[:loop_end] + main[loop_begin:loop_end] + main[last_begin:])       # nsub = 4
    check_hazards(main[:first_end] + main[last_begin:])                                  # nsub = 2
    check_hazards(g.out[main_end:])                      # the subroutines on their own (each starts with 32 wait states)
    return g, dict(first=first_end, loop=loop_end - loop_begin, last=main_end - last_begin, sub=len(g.out) - main_end)


def render(g):
    return "\n".join(ins.text for ins in g.out) + "\n"


def clobbers():
    c = [f"v{i}" for i in range(NV)] + [f"a{i}" for i in range(NA)] + [f"s{i}" for i in SGPRS] + ["vcc", "scc", "memory"]
    return ", ".join(f'"{x}"' for x in c)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "attn_fwd_asm.inc")
    parts = ["// GENERATED by gen_attn_asm.py -- do not edit; edit the generator and re-run it (transception_amd.build does).\n"]
    for half in ("bf16", "f16"):
        g, stats = generate(half)
        nm = sum(1 for i in g.out if i.kind == "mfma")
        parts.append(f"// {half}: {len(g.out)} lines, {nm} MFMAs; sections {stats}\n")
        parts.append(f"#define TC_ATTN_FWD_ASM_{half.upper()} R\"ASM(\n{render(g)})ASM\"\n")
    parts.append(f"#define TC_ATTN_FWD_ASM_CLOBBERS {clobbers()}\n")
    if TIMING:
        parts.append(f"#define TC_ATTN_ASM_TIMING {NSTAMP}\n")
    text = "".join(parts)
    if not os.path.exists(out) or open(out).read() != text:
        with open(out, "w") as f:
            f.write(text)
    if "-v" in sys.argv:
        print(text)


if __name__ == "__main__":
    main()
