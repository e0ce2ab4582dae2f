#!/usr/bin/env python3
"""Emits the hand-scheduled gfx950 instruction stream of the bridge SR-attention forward (csrc/attention_fwd_asm.hip).

Why a generator: on gfx950 the VALU work of one wave does not hide under the MFMAs of ANOTHER wave of the same SIMD (DESIGN.md,
scripts/exp/overlap.hip), only under MFMAs issued by the SAME wave in program order -- and hipcc would not keep such an order (rounds 2/3:
v60/v74/v80/v92 of scripts/exp/attn_exp.hip).  So the loop is written instruction by instruction: ONE wave per SIMD, each wave owns
three 32-query tiles, and every group of eight MFMAs (QK^T of one tile alternating with PV of another) carries the exp / row-sum /
pack arithmetic of the third tile plus the LDS fragment reads, in program order.  This file is that order, spelled as a small macro
assembler with two checks a human cannot be trusted with:
  * counted s_waitcnt lgkmcnt()/vmcnt() are inserted from a model of the in-order return queues, and the loop back-edge is
    verified to reproduce the queue state the loop body was generated for;
  * the MFMA / transcendental / permlane wait-state rules hipcc's hazard recogniser applies are re-checked on the dynamic
    instruction trace (inline asm gets no hazard handling from the compiler).

Reference semantics: softmax(Q K^T * scale) V of `M_EfficientSelfAtten` (networks/MSTr.py:2281-2287); the numerics (fixed integer
reference exponent, re-referenced when a 16-key partial row sum passes REREF) are those of attn_fwd_seg_kernel in attention_seg.hip.

    python gen_attn_asm.py            # writes attn_fwd_asm.inc next to this file
"""
from __future__ import annotations

import os
import sys

# ---------------------------------------------------------------------------------------------------------------------------------
# layout shared with attention_fwd_asm.hip (keep in sync with the constants there)
LDR_B = 144                     # bytes per LDS row: 64 halfs + 8 pad
SLOT_B = 16384                  # ring slot pitch (K sub-tile | V sub-tile), power of two so that the slot offset wraps with one s_and
VOFF_B = 32 * LDR_B             # V sub-tile inside a slot
NSLOT = 4
WT_B = 32 * LDR_B               # one 32-query wave tile
NEG_BIG = 0xF149F2CA            # -1.0e30f


def vreg(n, w=1):
    return f"v{n}" if w == 1 else f"v[{n}:{n + w - 1}]"


def areg(n, w=1):
    return f"a{n}" if w == 1 else f"a[{n}:{n + w - 1}]"


def regs(prefix, n, w=1):
    return [f"{prefix}{i}" for i in range(n, n + w)]


# register map -------------------------------------------------------------------------------------------------------------------
def S(t): return 16 * t                      # score tile of query tile t (16 VGPRs)
def NEGM(t): return 48 + 16 * t              # -m broadcast, the C operand of the first QK^T MFMA
E = 96                                       # exp scratch (16)
def P(t): return 112 + 8 * t                 # packed P (two B operands of 4)
def KF(ks): return 136 + 4 * ks              # K fragments of the current sub-tile
def VF(f): return 152 + 4 * f                # V^T fragments, f = 2 * k2 + blk
STK, STV = 168, 172
def L(t): return 176 + t
def M(t): return 179 + t
RS, T0, T1, T2, T3 = 182, 183, 184, 185, 186
KADDR, VADDR, WKADDR, WVADDR, GK, GV = 187, 188, 189, 190, 191, 192
KBASE, VBASE, WKBASE, WVBASE, QADDR, OADDR, LSEADDR, MASK, NEGBIGR, T4, T5 = 193, 194, 195, 196, 197, 198, 199, 200, 201, 202, 203
NV = 204
def O(t, blk): return 32 * t + 16 * blk      # AGPR
def QF(t, ks): return 96 + 16 * t + 4 * ks   # AGPR
NA = 144 + (48 if 'ones' in os.environ.get('TC_ATTN_ABLATE', '') else 0)
S_CNT, S_THR, S_SV, S_SK, S_SW, S_RA, S_LN2, S_STEPK, S_STEPV, S_T = 60, 61, 62, 63, 64, 66, 68, 69, 70, 71
SGPRS = list(range(60, 72))
TIMING = bool(int(os.environ.get("TC_ATTN_TIMING", "0")))       # experiment builds: s_memtime stamps at section boundaries
NSTAMP = 6
ABLATE = os.environ.get("TC_ATTN_ABLATE", "")                    # timing experiments only (wrong results): noexp, novalu, nomfma, nostage, nobar
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
            n = len(q) - idx - 1
            self._push(Ins(f"s_waitcnt {name}({n})", "wait"))
            del q[: idx + 1]

    def wait_lgkm(self, n):
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
    def mfma(self, d, a, b, c, acc_d=False, a_acc=False, b_acc=False, c_zero=False):
        dn = regs("a" if acc_d else "v", d, 16)
        an = regs("a" if a_acc else "v", a, 4)
        bn = regs("a" if b_acc else "v", b, 4)
        cn = [] if c_zero else regs("a" if acc_d else "v", c, 16)
        ds = (areg if acc_d else vreg)(d, 16)
        cs = "0" if c_zero else (areg if acc_d else vreg)(c, 16)
        if "nomfma" in ABLATE:
            return
        self.emit(f"v_mfma_f32_32x32x16_{self.half} {ds}, {(areg if a_acc else vreg)(a, 4)}, {(areg if b_acc else vreg)(b, 4)}, {cs}",
                  "mfma", an + bn + cn, dn)

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
def mf_qk(g, t, ks):
    g.mfma(S(t), KF(ks), QF(t, ks), NEGM(t) if ks == 0 else S(t), b_acc=True)


def mf_pv(g, t, f):
    k2, blk = f >> 1, f & 1
    g.mfma(O(t, blk), VF(f), P(t) + 4 * k2, O(t, blk), acc_d=True)
    if "ones" in ABLATE and blk == 1:
        g.mfma(144 + 16 * t, VF(f), P(t) + 4 * k2, 144 + 16 * t, acc_d=True)


def vf_load(g, f):
    k2, blk = f >> 1, f & 1
    for e in range(2):
        g.emit(f"ds_read_b64_tr_b16 {vreg(VF(f) + 2 * e, 2)}, {vreg(VADDR)} offset:{(2 * k2 + e) * LDR_B + 64 * blk}", "ds_read",
               [vreg(VADDR)], regs("v", VF(f) + 2 * e, 2))


def kf_load(g, ks, addr=KADDR):
    g.emit(f"ds_read_b128 {vreg(KF(ks), 4)}, {vreg(addr)} offset:{32 * ks}", "ds_read", [vreg(addr)], regs("v", KF(ks), 4))


def cvt_name(half):
    return "v_cvt_pk_bf16_f32" if half == "bf16" else "v_cvt_pk_f16_f32"


def sm_items(g, t, mode):
    """The softmax filler list of query tile t as closures (each emits one instruction): exp -> E, row sum -> RS, re-reference check,
    L += RS, pack -> P(t).  mode: 'first' (the reference exponent is set here), 'loop', 'last' (keys past Nk masked first)."""
    it = []
    if "novalu" in ABLATE and mode == "loop":
        return it
    if mode == "last":
        for r in range(16):
            it.append(lambda r=r: g.valu(f"v_bfe_i32 {vreg(T4)}, {vreg(MASK)}, {r}, 1", [vreg(MASK)], [vreg(T4)]))
            it.append(lambda r=r: g.valu(f"v_bfi_b32 {vreg(S(t) + r)}, {vreg(T4)}, {vreg(NEGBIGR)}, {vreg(S(t) + r)}",
                                         [vreg(T4), vreg(NEGBIGR), vreg(S(t) + r)], [vreg(S(t) + r)]))
    if mode == "first":
        it.append(lambda: g.emit(f"s_call_b64 s[{S_RA}:{S_RA + 1}], .Lfirst{t}_%=", "call"))
    else:
        def ex(r): return lambda: g.valu(f"v_exp_f32_e32 {vreg(E + r)}, {vreg(S(t) + r)}", [vreg(S(t) + r)], [vreg(E + r)], trans=True)
        def ad(r): return lambda: g.valu(f"v_add_f32_e32 {vreg(RS)}, {vreg(RS)}, {vreg(E + r)}", [vreg(RS), vreg(E + r)], [vreg(RS)])
        if "ones" in ABLATE and mode == "loop":      # timing experiment: row sums by two extra MFMAs, overflow check = max3 tree on S
            s0 = S(t)
            it.append(lambda: g.valu(f"v_max3_f32 {vreg(RS)}, {vreg(s0)}, {vreg(s0 + 1)}, {vreg(s0 + 2)}", regs("v", s0, 3), [vreg(RS)]))
            for r in range(3, 15, 2):
                it.append(lambda r=r: g.valu(f"v_max3_f32 {vreg(RS)}, {vreg(RS)}, {vreg(s0 + r)}, {vreg(s0 + r + 1)}", [vreg(RS)] + regs("v", s0 + r, 2), [vreg(RS)]))
            it.append(lambda: g.valu(f"v_max_f32_e32 {vreg(RS)}, {vreg(RS)}, {vreg(s0 + 15)}", [vreg(RS), vreg(s0 + 15)], [vreg(RS)]))
            it += [ex(r) for r in range(16)]
        elif "pkadd" in ABLATE and mode == "loop":   # timing experiment: packed fp32 adds
            it += [ex(0), ex(1), ex(2), ex(3)]
            for r in range(4, 16, 2):
                it += [ex(r), ex(r + 1)]
                a = E + r - 4 if r > 4 else E
                it.append(lambda r=r: g.valu(f"v_pk_add_f32 {vreg(E + r - 2, 2)}, {vreg(E + r - 4, 2)}, {vreg(E + r - 2, 2)}", regs("v", E + r - 4, 4), regs("v", E + r - 2, 2)))
            it.append(lambda: g.valu(f"v_pk_add_f32 {vreg(E + 14, 2)}, {vreg(E + 12, 2)}, {vreg(E + 14, 2)}", regs("v", E + 12, 4), regs("v", E + 14, 2)))
            it.append(lambda: g.valu(f"v_add_f32_e32 {vreg(RS)}, {vreg(E + 14)}, {vreg(E + 15)}", regs("v", E + 14, 2), [vreg(RS)]))
        else:
            it += [ex(0), ex(1), ex(2)]
            it.append(lambda: g.valu(f"v_add_f32_e32 {vreg(RS)}, {vreg(E)}, {vreg(E + 1)}", [vreg(E), vreg(E + 1)], [vreg(RS)]))
            for r in range(3, 16):
                it += [ex(r), ad(r - 1)]
            it.append(ad(15))
        def check():                           # one unit: nothing may be scheduled between the branch and its target
            g.valu(f"v_cmp_ngt_f32_e32 vcc, s{S_THR}, {vreg(RS)}", [vreg(RS)], ["vcc"])
            g.emit(f"s_cbranch_vccz .Lskip{g.uid}_%=", "branch", ["vcc"])
            g.emit(f"s_call_b64 s[{S_RA}:{S_RA + 1}], .Lslow{t}_%=", "call")
            g.label(f".Lskip{g.uid}_%=")
            g.uid += 1
        it.append(check)
    it.append(lambda: g.valu(f"v_add_f32_e32 {vreg(L(t))}, {vreg(L(t))}, {vreg(RS)}", [vreg(L(t)), vreg(RS)], [vreg(L(t))]))
    for q in range(8):
        it.append(lambda q=q: g.valu(f"{cvt_name(g.half)} {vreg(P(t) + q)}, {vreg(E + 2 * q)}, {vreg(E + 2 * q + 1)}",
                                     [vreg(E + 2 * q), vreg(E + 2 * q + 1)], [vreg(P(t) + q)]))
    return it


def pair_slot(g, mfmas, anchored, free, head=()):
    """mfmas: list of closures; anchored[k]: closures placed right after MFMA k; free: ordered closures spread over the gaps;
    head: closures placed before the first MFMA."""
    for f in head:
        f()
    n = len(mfmas)
    per = [len(free) // n + (1 if k < len(free) % n else 0) for k in range(n)]
    pos = 0
    for k, m in enumerate(mfmas):
        m()
        for f in anchored.get(k, ()):
            f()
        for f in free[pos: pos + per[k]]:
            f()
        pos += per[k]


def body(g, mode):
    """One key sub-tile i of all three query tiles: Y (QK_1(i) | PV_2(i-1), softmax_0(i)), Z (QK_2(i) | PV_0(i), softmax_1(i)),
    X' (QK_0(i+1) | PV_1(i), softmax_2(i)).  mode 'first': no sub-tile i-1; 'last': no sub-tile i+1, masked keys."""
    first, last = mode == "first", mode == "last"
    # ---- Y
    head = []
    if not first and not last:
        head.append(lambda: g.wait_lgkm(0))
        if "nobar" not in ABLATE:
            head.append(lambda: g.emit("s_barrier", "barrier"))
    head.append(lambda: g.valu(f"v_add_u32_e32 {vreg(VADDR)}, s{S_SV}, {vreg(VBASE)}", [vreg(VBASE)], [vreg(VADDR)]))
    if not last and not ("nostage" in ABLATE and mode == "loop"):
        head.append(lambda: g.emit(f"buffer_load_dwordx4 {vreg(STK, 4)}, {vreg(GK)}, %10, 0 offen", "vmem_load", [vreg(GK)], regs("v", STK, 4)))
        head.append(lambda: g.emit(f"buffer_load_dwordx4 {vreg(STV, 4)}, {vreg(GV)}, %11, 0 offen", "vmem_load", [vreg(GV)], regs("v", STV, 4)))
        head.append(lambda: g.valu(f"v_add_u32_e32 {vreg(GK)}, s{S_STEPK}, {vreg(GK)}", [vreg(GK)], [vreg(GK)]))
        head.append(lambda: g.valu(f"v_add_u32_e32 {vreg(GV)}, s{S_STEPV}, {vreg(GV)}", [vreg(GV)], [vreg(GV)]))
    if first:
        mf = [lambda ks=ks: mf_qk(g, 1, ks) for ks in range(4)]
        anch = {k: [lambda f=k: vf_load(g, f)] for k in range(4)}
    else:
        mf, anch = [], {}
        for ks in range(4):
            mf.append(lambda ks=ks: mf_qk(g, 1, ks))
            mf.append(lambda f=ks: mf_pv(g, 2, f))
            anch[2 * ks + 1] = [lambda f=ks: vf_load(g, f)]
    pair_slot(g, mf, anch, sm_items(g, 0, mode), head)
    # ---- Z
    head = []
    if not last:
        head.append(lambda: g.valu(f"v_add_u32_e32 {vreg(KADDR)}, s{S_SK}, {vreg(KBASE)}", [vreg(KBASE)], [vreg(KADDR)]))
    mf, anch = [], {}
    for ks in range(4):
        mf.append(lambda ks=ks: mf_qk(g, 2, ks))
        mf.append(lambda f=ks: mf_pv(g, 0, f))
        if not last:
            anch[2 * ks] = [lambda ks=ks: kf_load(g, ks)]
    pair_slot(g, mf, anch, sm_items(g, 1, mode), head)
    # ---- X'
    mf, anch = [], {}
    for ks in range(4):
        if not last:
            mf.append(lambda ks=ks: mf_qk(g, 0, ks))
        mf.append(lambda f=ks: mf_pv(g, 1, f))
    if not last and not ("nostage" in ABLATE and mode == "loop"):
        def stash():
            g.wait_vm(0)
            g.valu(f"v_add_u32_e32 {vreg(WKADDR)}, s{S_SW}, {vreg(WKBASE)}", [vreg(WKBASE)], [vreg(WKADDR)])
            g.valu(f"v_add_u32_e32 {vreg(WVADDR)}, s{S_SW}, {vreg(WVBASE)}", [vreg(WVBASE)], [vreg(WVADDR)])
            g.emit(f"ds_write_b128 {vreg(WKADDR)}, {vreg(STK, 4)}", "ds_write", [vreg(WKADDR)] + regs("v", STK, 4))
            g.emit(f"ds_write_b128 {vreg(WVADDR)}, {vreg(STV, 4)}", "ds_write", [vreg(WVADDR)] + regs("v", STV, 4))
            for s in (S_SV, S_SK, S_SW):      # ring slot offsets of the next sub-tile
                g.salu(f"s_add_u32 s{s}, s{s}, {SLOT_B}")
                g.salu(f"s_and_b32 s{s}, s{s}, {SLOT_B * NSLOT - 1}")
        anch[2] = [stash]
    pair_slot(g, mf, anch, sm_items(g, 2, mode))


def subroutine(g, t, first):
    """Sets (first) or raises (loop) the reference exponent of query tile t from the score tile S(t) = s * qs - m_old, rescales
    what was accumulated against the old one, and recomputes E / RS.  Called, never fallen into; returns through s[S_RA:S_RA+1]."""
    g.label(f".L{'first' if first else 'slow'}{t}_%=")
    g.nop(32)                                  # every MFMA issued before the call has retired (S(t), O(t) quiescent)
    s = S(t)
    g.valu(f"v_max3_f32 {vreg(T0)}, {vreg(s)}, {vreg(s + 1)}, {vreg(s + 2)}", regs("v", s, 3), [vreg(T0)])
    for r in range(3, 15, 2):
        g.valu(f"v_max3_f32 {vreg(T0)}, {vreg(T0)}, {vreg(s + r)}, {vreg(s + r + 1)}", [vreg(T0)] + regs("v", s + r, 2), [vreg(T0)])
    g.valu(f"v_max_f32_e32 {vreg(T0)}, {vreg(T0)}, {vreg(s + 15)}", [vreg(T0), vreg(s + 15)], [vreg(T0)])
    g.valu(f"v_mov_b32_e32 {vreg(T1)}, {vreg(T0)}", [vreg(T0)], [vreg(T1)])
    g.nop(2)
    g.emit(f"v_permlane32_swap_b32_e32 {vreg(T0)}, {vreg(T1)}", "permlane", [vreg(T0), vreg(T1)], [vreg(T0), vreg(T1)])
    g.valu(f"v_max_f32_e32 {vreg(T0)}, {vreg(T0)}, {vreg(T1)}", [vreg(T0), vreg(T1)], [vreg(T0)])
    g.valu(f"v_ceil_f32_e32 {vreg(T2)}, {vreg(T0)}", [vreg(T0)], [vreg(T2)])               # T2 = m_delta
    if not first:
        g.valu(f"v_max_f32_e32 {vreg(T2)}, 0, {vreg(T2)}", [vreg(T2)], [vreg(T2)])
        g.valu(f"v_sub_f32_e32 {vreg(T1)}, 0, {vreg(T2)}", [vreg(T2)], [vreg(T1)])
        g.valu(f"v_exp_f32_e32 {vreg(T3)}, {vreg(T1)}", [vreg(T1)], [vreg(T3)], trans=True)  # T3 = alpha
        g.nop(1)
        g.valu(f"v_mul_f32_e32 {vreg(L(t))}, {vreg(L(t))}, {vreg(T3)}", [vreg(L(t)), vreg(T3)], [vreg(L(t))])
        for blk in range(2):
            for r in range(16):
                a = O(t, blk) + r
                g.valu(f"v_accvgpr_read_b32 {vreg(T1)}, {areg(a)}", [areg(a)], [vreg(T1)])
                g.valu(f"v_mul_f32_e32 {vreg(T1)}, {vreg(T1)}, {vreg(T3)}", [vreg(T1), vreg(T3)], [vreg(T1)])
                g.valu(f"v_accvgpr_write_b32 {areg(a)}, {vreg(T1)}", [vreg(T1)], [areg(a)])
    g.valu(f"v_add_f32_e32 {vreg(M(t))}, {vreg(M(t))}, {vreg(T2)}", [vreg(M(t)), vreg(T2)], [vreg(M(t))])
    for r in range(16):
        g.valu(f"v_sub_f32_e32 {vreg(NEGM(t) + r)}, {vreg(NEGM(t) + r)}, {vreg(T2)}", [vreg(NEGM(t) + r), vreg(T2)], [vreg(NEGM(t) + r)])
    for r in range(16):
        g.valu(f"v_sub_f32_e32 {vreg(s + r)}, {vreg(s + r)}, {vreg(T2)}", [vreg(s + r), vreg(T2)], [vreg(s + r)])
    for r in range(16):
        g.valu(f"v_exp_f32_e32 {vreg(E + r)}, {vreg(s + r)}", [vreg(s + r)], [vreg(E + r)], trans=True)
    g.nop(1)
    g.valu(f"v_add_f32_e32 {vreg(RS)}, {vreg(E)}, {vreg(E + 1)}", [vreg(E), vreg(E + 1)], [vreg(RS)])
    for r in range(2, 16):
        g.valu(f"v_add_f32_e32 {vreg(RS)}, {vreg(RS)}, {vreg(E + r)}", [vreg(RS), vreg(E + r)], [vreg(RS)])
    g.nop(4)
    g.emit(f"s_setpc_b64 s[{S_RA}:{S_RA + 1}]", "ret")


def prologue(g):
    ins = [(KBASE, 0), (VBASE, 1), (WKBASE, 2), (WVBASE, 3), (GK, 4), (GV, 5), (QADDR, 6), (OADDR, 7), (LSEADDR, 8), (MASK, 9)]
    for r, k in ins:
        g.valu(f"v_mov_b32_e32 {vreg(r)}, %{k}", [], [vreg(r)])
    g.salu(f"s_mov_b32 s{S_CNT}, %12")
    g.salu(f"s_mov_b32 s{S_STEPK}, %13")
    g.salu(f"s_mov_b32 s{S_STEPV}, %14")
    g.salu(f"s_mov_b32 s{S_THR}, {'0x4e800000' if g.half == 'bf16' else '0x46800000'}")    # REREF: 2^30 (bf16 P) / 2^14 (fp16 P)
    g.salu(f"s_mov_b32 s{S_LN2}, 0x3f317218")
    g.salu(f"s_mov_b32 s{S_SV}, 0")
    g.salu(f"s_mov_b32 s{S_SK}, {SLOT_B}")
    g.salu(f"s_mov_b32 s{S_SW}, {3 * SLOT_B}")
    g.salu(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 2")                                           # steady iterations: sub-tiles 1 .. nsub - 2
    for t in range(3):
        for ks in range(4):
            g.emit(f"ds_read_b128 {areg(QF(t, ks), 4)}, {vreg(QADDR)} offset:{t * WT_B + 32 * ks}", "ds_read", [vreg(QADDR)], regs("a", QF(t, ks), 4))
    for ks in range(4):
        kf_load(g, ks, KBASE)                                                             # K(0): slot 0
    g.valu(f"v_mov_b32_e32 {vreg(NEGBIGR)}, 0x{NEG_BIG:08x}", [], [vreg(NEGBIGR)])
    for t in range(3):
        g.valu(f"v_mov_b32_e32 {vreg(L(t))}, 0", [], [vreg(L(t))])
        g.valu(f"v_mov_b32_e32 {vreg(M(t))}, 0", [], [vreg(M(t))])
        for r in range(16):
            g.valu(f"v_mov_b32_e32 {vreg(NEGM(t) + r)}, 0", [], [vreg(NEGM(t) + r)])
        for r in range(32):
            g.valu(f"v_accvgpr_write_b32 {areg(32 * t + r)}, 0", [], [areg(32 * t + r)])
    for ks in range(4):                                                                   # X(0): QK_0(0)
        mf_qk(g, 0, ks)


def epilogue(g):
    g.nop(32)
    for t in range(3):
        g.valu(f"v_mov_b32_e32 {vreg(T0)}, {vreg(L(t))}", [vreg(L(t))], [vreg(T0)])
        g.valu(f"v_mov_b32_e32 {vreg(T1)}, {vreg(L(t))}", [vreg(L(t))], [vreg(T1)])
        g.nop(2)
        g.emit(f"v_permlane32_swap_b32_e32 {vreg(T0)}, {vreg(T1)}", "permlane", [vreg(T0), vreg(T1)], [vreg(T0), vreg(T1)])
        g.valu(f"v_add_f32_e32 {vreg(T0)}, {vreg(T0)}, {vreg(T1)}", [vreg(T0), vreg(T1)], [vreg(T0)])          # T0 = row sum over both halves
        g.valu(f"v_rcp_f32_e32 {vreg(T5)}, {vreg(T0)}", [vreg(T0)], [vreg(T5)], trans=True)
        g.valu(f"v_log_f32_e32 {vreg(T1)}, {vreg(T0)}", [vreg(T0)], [vreg(T1)], trans=True)
        g.nop(1)
        g.valu(f"v_add_f32_e32 {vreg(T1)}, {vreg(T1)}, {vreg(M(t))}", [vreg(T1), vreg(M(t))], [vreg(T1)])
        g.valu(f"v_mul_f32_e32 {vreg(T1)}, s{S_LN2}, {vreg(T1)}", [vreg(T1)], [vreg(T1)])
        g.emit(f"ds_write_b32 {vreg(LSEADDR)}, {vreg(T1)} offset:{128 * t}", "ds_write", [vreg(LSEADDR), vreg(T1)])
        for blk in range(2):
            for q in range(4):
                for e in range(4):
                    a = O(t, blk) + 4 * q + e
                    g.valu(f"v_accvgpr_read_b32 {vreg(E + e)}, {areg(a)}", [areg(a)], [vreg(E + e)])
                for e in range(4):
                    g.valu(f"v_mul_f32_e32 {vreg(E + e)}, {vreg(E + e)}, {vreg(T5)}", [vreg(E + e), vreg(T5)], [vreg(E + e)])
                g.valu(f"{cvt_name(g.half)} {vreg(E + 4)}, {vreg(E)}, {vreg(E + 1)}", regs("v", E, 2), [vreg(E + 4)])
                g.valu(f"{cvt_name(g.half)} {vreg(E + 5)}, {vreg(E + 2)}, {vreg(E + 3)}", regs("v", E + 2, 2), [vreg(E + 5)])
                g.emit(f"ds_write_b64 {vreg(OADDR)}, {vreg(E + 4, 2)} offset:{t * WT_B + 64 * blk + 16 * q}", "ds_write", [vreg(OADDR), vreg(E + 4), vreg(E + 5)])
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
    g.stamp(1)
    pro_end = len(g.out)
    body(g, "first")
    g.stamp(2)
    first_end = len(g.out)
    st_in = g.state()
    g.salu(f"s_cmp_lt_i32 s{S_CNT}, 1")
    g.emit("s_cbranch_scc1 .Llast_%=", "branch")
    g.label(".Lloop_%=")
    loop_begin = len(g.out)
    body(g, "loop")
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
    g.nop(11)                                            # entered from the loop or straight from the first body (nsub = 2)
    body(g, "last")
    for f in range(4):                                   # drain: PV_2(last)
        mf_pv(g, 2, f)
    g.stamp(4)
    epilogue(g)
    g.stamp(5)
    if TIMING:
        for k in range(NSTAMP):
            g.valu(f"v_mov_b32_e32 {vreg(E)}, s{72 + 2 * k}", [], [vreg(E)])
            g.valu(f"v_mov_b32_e32 {vreg(E + 1)}, s{73 + 2 * k}", [], [vreg(E + 1)])
            g.emit(f"ds_write_b64 {vreg(LSEADDR)}, {vreg(E, 2)} offset:{4096 + 8 * k}", "ds_write", [vreg(LSEADDR), vreg(E), vreg(E + 1)])
        g.wait_lgkm(0)
    g.emit("s_branch .Lend_%=", "branch")
    main_end = len(g.out)
    for t in range(3):
        subroutine(g, t, True)
        subroutine(g, t, False)
    g.label(".Lend_%=")
    # dynamic trace for nsub = 4: prologue, first, loop, loop, last ...
    main = g.out[:main_end]
    trace = main[:loop_end] + main[loop_begin:loop_end] + main[last_begin:]
    check_hazards(trace)
    check_hazards(main[:first_end] + main[last_begin:])  # nsub = 2
    sub = g.out[main_end:]                               # the subroutines on their own (each starts with 32 wait states)
    check_hazards(sub)
    return g, dict(pro=pro_end, first=first_end - pro_end, loop=loop_end - loop_begin, last=main_end - last_begin, sub=len(sub))


def render(g):
    lines = []
    for ins in g.out:
        lines.append(ins.text)
    return "\n".join(lines) + "\n"


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
