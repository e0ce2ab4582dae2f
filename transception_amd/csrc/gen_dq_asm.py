#!/usr/bin/env python3
"""Emits the hand-scheduled gfx950 instruction stream of the bridge SR-attention dQ kernel (attn_bwd_dq_asm_kernel in attention_seg.hip)
with the macro assembler of gen_attn_asm.py.

The forward's decomposition and ring: one 12-wave workgroup per CU, a 32-query tile per wave (Q / dO fragments and the dQ^T accumulators
stay in AGPRs), 32-key K | V sub-tiles through the 12-slot LDS ring (waves 0-3 stage K, 4-7 V, buffer loads six sub-tiles ahead, a
barrier per four sub-tiles).  K is stored ONCE, rows in key_row order: 16-byte fragment reads for S^T = K Q^T and transpose reads for
dQ^T += K^T dS^T.  Three waves per SIMD leave 84 + 84 registers, so ONE pool of four 4-register fragments serves the three fragment
kinds in turn -- MFMA i of a group reads fragment i and the load behind it refills it for MFMA i of the next group:

  iteration j:  dQ^T(j)       4 MFMAs on the K^T fragments, carrying exp2 / subtract / multiply of sub-tile j+1; refill: K rows of j+2
                S^T(j+2)      4 MFMAs (C = -lse log2 e: P = exp2(S) directly), carrying pack dS(j+1); refill: V rows of j+2
                dP^T(j+2)     4 MFMAs; refill: K^T fragments of j+1; ring store of sub-tile j+6, bookkeeping

Arithmetic = attn_bwd_dq_seg_kernel<H, 1> (Q stored as q * scale * log2 e): dQ = scale * (P (dP - delta)) K.

    python gen_dq_asm.py            # writes attn_dq_asm.inc next to this file
"""
from __future__ import annotations

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_attn_asm import AHEAD, LDR_B, NSLOT, PERIOD, SLOT_B, Gen, Ins, areg, check_hazards, cvt_name, regs, vreg  # noqa: E402

VOFF = 32 * LDR_B                 # V sub-tile behind the K sub-tile of a slot

# VGPRs
S, DP, NEGL = 0, 16, 32
PD = 48                           # packed dS: two B operands of 4
ST = 56
T0, T1 = 60, 61
AK, AV, AT, AW = 62, 63, 64, 65
NV = 66
# AGPRs
def DQ(blk): return 16 * blk
def QF(ks): return 32 + 4 * ks
def GF(ks): return 48 + 4 * ks
def F(i): return 64 + 4 * i
NA = 80
# operands: %0 "+v" global offset of this thread's staging chunk
(OP_GOFF, OP_KBASE, OP_VBASE, OP_TBASE, OP_WBASE, OP_QOFF, OP_GOOFF, OP_OADDR, OP_NL2, OP_DLT, OP_MASK,
 OP_RSRC, OP_RQ, OP_RG, OP_NSUB, OP_STEP, OP_WEXEC, OP_SCALE) = (f"%{i}" for i in range(18))
S_CNT, S_SV, S_SK, S_SW, S_PH, S_STEP, S_SCL = 60, 62, 63, 64, 65, 69, 68
SGPRS = list(range(60, 70))


def krow_load(g, i, base):
    b = base if isinstance(base, str) else vreg(base)
    g.emit(f"ds_read_b128 {areg(F(i), 4)}, {b} offset:{32 * i}", "ds_read", [b], regs("a", F(i), 4))


def tr_load(g, i, base):
    """K^T fragment i = 2 k2 + blk: keys 16 h + 8 k2 .. + 7 (two 4-key transpose reads), d block blk."""
    k2, blk = i >> 1, i & 1
    for e in range(2):
        g.emit(f"ds_read_b64_tr_b16 {areg(F(i) + 2 * e, 2)}, {vreg(base)} offset:{(2 * k2 + e) * LDR_B + 64 * blk}", "ds_read",
               [vreg(base)], regs("a", F(i) + 2 * e, 2))


def mf_dq(g, i):
    k2, blk = i >> 1, i & 1
    g.mfma(DQ(blk), F(i), PD + 4 * k2, DQ(blk), acc_d=True)


def mf_s(g, ks):
    g.mfma(S, F(ks), QF(ks), NEGL if ks == 0 else S, b_acc=True)


def mf_dp(g, ks):
    if ks == 0:       # C = 0 (inline constant)
        g.emit(f"v_mfma_f32_32x32x16_{g.half} {vreg(DP, 16)}, {areg(F(0), 4)}, {areg(GF(0), 4)}, 0", "mfma",
               regs("a", F(0), 4) + regs("a", GF(0), 4), regs("v", DP, 16))
    else:
        g.mfma(DP, F(ks), GF(ks), DP, b_acc=True)


def valu_items(g, masked):
    """P = exp2(S) in place (keys past Nk: 0), dS = P (dP - delta) in place of dP."""
    it = [lambda r=r: g.valu(f"v_exp_f32_e32 {vreg(S + r)}, {vreg(S + r)}", [vreg(S + r)], [vreg(S + r)], trans=True) for r in range(16)]
    if masked:
        for r in range(16):
            it.append(lambda r=r: g.valu(f"v_bfe_i32 {vreg(T0)}, {OP_MASK}, {r}, 1", [], [vreg(T0)]))
            it.append(lambda r=r: g.valu(f"v_bfi_b32 {vreg(S + r)}, {vreg(T0)}, 0, {vreg(S + r)}", [vreg(T0), vreg(S + r)], [vreg(S + r)]))
    it += [lambda r=r: g.valu(f"v_sub_f32_e32 {vreg(DP + r)}, {vreg(DP + r)}, {OP_DLT}", [vreg(DP + r)], [vreg(DP + r)]) for r in range(16)]
    it += [lambda r=r: g.valu(f"v_mul_f32_e32 {vreg(DP + r)}, {vreg(DP + r)}, {vreg(S + r)}", [vreg(DP + r), vreg(S + r)], [vreg(DP + r)]) for r in range(16)]
    return it


def pack_items(g):
    return [lambda q=q: g.valu(f"{cvt_name(g.half)} {vreg(PD + q)}, {vreg(DP + 2 * q)}, {vreg(DP + 2 * q + 1)}", regs("v", DP + 2 * q, 2), [vreg(PD + q)])
            for q in range(8)]


def interleave(g, n, mf, after, free, head=()):
    for f in head:
        f()
    per = [len(free) // n + (1 if k < len(free) % n else 0) for k in range(n)]
    pos = 0
    for i in range(n):
        if mf is not None:
            mf(i)
        after(i)
        for f in free[pos: pos + per[i]]:
            f()
        pos += per[i]


def iteration(g, mode):
    first, last, drain = mode == "first", mode == "last", mode == "drain"
    stage = not (last or drain)
    nothing = lambda i: None
    # ---- dQ^T(j) | VALU(j+1) | K rows (j+2)
    head = []
    if mode == "loop":
        def bar():
            g.wait_lgkm(0)
            g.salu(f"s_cmp_lg_u32 s{S_PH}, 0")
            g.emit(f"s_cbranch_scc1 .Lnobar{g.uid}_%=", "branch")
            g.emit("s_barrier", "barrier")
            g.label(f".Lnobar{g.uid}_%=")
            g.uid += 1
            g.salu(f"s_add_u32 s{S_PH}, s{S_PH}, 1")
            g.salu(f"s_and_b32 s{S_PH}, s{S_PH}, {PERIOD - 1}")
        head.append(bar)
    if stage:
        head.append(lambda: g.salu(f"s_mov_b64 exec, {OP_WEXEC}"))
        head.append(lambda: g.emit(f"buffer_load_dwordx4 {vreg(ST, 4)}, {OP_GOFF}, {OP_RSRC}, 0 offen", "vmem_load", [], regs("v", ST, 4)))
        head.append(lambda: g.salu("s_mov_b64 exec, -1"))
        head.append(lambda: g.valu(f"v_add_u32_e32 {OP_GOFF}, s{S_STEP}, {OP_GOFF}", [], []))
        head.append(lambda: g.valu(f"v_add_u32_e32 {vreg(AK)}, s{S_SK}, {OP_KBASE}", [], [vreg(AK)]))
    if not first:
        head.append(lambda: g.salu("s_setprio 1"))
    interleave(g, 4, None if first else (lambda i: mf_dq(g, i)), (lambda i: krow_load(g, i, AK)) if stage else nothing,
               [] if drain else valu_items(g, last), head)
    if not first:
        g.salu("s_setprio 0")
    if drain:
        return
    # ---- S^T(j+2) | V rows (j+2) | pack dS(j+1): the packs read the dS registers the dP^T group overwrites, not the S registers this one
    #      writes, and sit between a V-row refill and the dP^T MFMA that needs it
    head = []
    if not last:
        head.append(lambda: g.valu(f"v_add_u32_e32 {vreg(AV)}, s{S_SK}, {OP_VBASE}", [], [vreg(AV)]))
    head.append(lambda: g.valu(f"v_add_u32_e32 {vreg(AT)}, s{S_SV}, {OP_TBASE}", [], [vreg(AT)]))
    interleave(g, 4, None if last else (lambda i: mf_s(g, i)), (lambda i: krow_load(g, i, AV)) if not last else nothing, pack_items(g), head)
    # ---- dP^T(j+2) | K^T fragments (j+1) | ring store, slot bookkeeping
    free = []
    if stage:
        def stash():
            g.wait_vm(0)
            g.valu(f"v_add_u32_e32 {vreg(AW)}, s{S_SW}, {OP_WBASE}", [], [vreg(AW)])
            g.salu(f"s_mov_b64 exec, {OP_WEXEC}")
            g.emit(f"ds_write_b128 {vreg(AW)}, {vreg(ST, 4)}", "ds_write", [vreg(AW)] + regs("v", ST, 4))
            g.salu("s_mov_b64 exec, -1")
        free.append(stash)
    if not last:
        def slots():
            for s in (S_SV, S_SK, S_SW):
                g.salu(f"s_add_u32 s{s}, s{s}, {SLOT_B}")
                g.salu(f"s_cmp_eq_u32 s{s}, {SLOT_B * NSLOT}")
                g.salu(f"s_cselect_b32 s{s}, 0, s{s}")
        free.append(slots)
    noop = lambda: None
    free = [noop] * (4 - len(free)) + free
    interleave(g, 4, None if last else (lambda i: mf_dp(g, i)), lambda i: tr_load(g, i, AT), free)


def prologue(g):
    g.salu(f"s_mov_b32 s{S_CNT}, {OP_NSUB}")
    g.salu(f"s_mov_b32 s{S_STEP}, {OP_STEP}")
    g.salu(f"s_mov_b32 s{S_SCL}, {OP_SCALE}")
    g.salu(f"s_mov_b32 s{S_SV}, 0")                                                       # K^T fragments of sub-tile 0
    g.salu(f"s_mov_b32 s{S_SK}, {SLOT_B}")                                                # K / V rows of sub-tile 1
    g.salu(f"s_mov_b32 s{S_SW}, {(AHEAD - 1) * SLOT_B}")
    g.salu(f"s_mov_b32 s{S_PH}, {2 % PERIOD}")
    g.salu(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 2")
    for ks in range(4):                                                                   # Q / dO fragments of the wave's 32 queries: global -> AGPRs
        g.emit(f"buffer_load_dwordx4 {areg(QF(ks), 4)}, {OP_QOFF}, {OP_RQ}, 0 offen offset:{32 * ks}", "vmem_load", [], regs("a", QF(ks), 4))
        g.emit(f"buffer_load_dwordx4 {areg(GF(ks), 4)}, {OP_GOOFF}, {OP_RG}, 0 offen offset:{32 * ks}", "vmem_load", [], regs("a", GF(ks), 4))
    for i in range(4):
        krow_load(g, i, OP_KBASE)                                                         # K rows of sub-tile 0 (slot 0)
    for r in range(16):
        g.valu(f"v_mov_b32_e32 {vreg(NEGL + r)}, {OP_NL2}", [], [vreg(NEGL + r)])
    for r in range(32):
        g.valu(f"v_accvgpr_write_b32 {areg(r)}, 0", [], [areg(r)])
    g.wait_vm(0)
    for i in range(4):
        mf_s(g, i)
        g.emit(f"ds_read_b128 {areg(F(i), 4)}, {OP_VBASE} offset:{32 * i}", "ds_read", [], regs("a", F(i), 4))
    for i in range(4):
        mf_dp(g, i)


def epilogue(g):
    g.nop(32)
    for blk in range(2):
        for q in range(4):
            for e in range(4):
                a = DQ(blk) + 4 * q + e
                g.valu(f"v_accvgpr_read_b32 {vreg(S + e)}, {areg(a)}", [areg(a)], [vreg(S + e)])
            for e in range(4):
                g.valu(f"v_mul_f32_e32 {vreg(S + e)}, s{S_SCL}, {vreg(S + e)}", [vreg(S + e)], [vreg(S + e)])
            g.valu(f"{cvt_name(g.half)} {vreg(S + 4)}, {vreg(S)}, {vreg(S + 1)}", regs("v", S, 2), [vreg(S + 4)])
            g.valu(f"{cvt_name(g.half)} {vreg(S + 5)}, {vreg(S + 2)}, {vreg(S + 3)}", regs("v", S + 2, 2), [vreg(S + 5)])
            g.emit(f"ds_write_b64 {OP_OADDR}, {vreg(S + 4, 2)} offset:{64 * blk + 16 * q}", "ds_write", [vreg(S + 4), vreg(S + 5)])
    g.wait_lgkm(0)


def generate(half):
    g = Gen(half)
    prologue(g)
    g.wait_lgkm(0)
    iteration(g, "first")
    first_end = len(g.out)
    st_in = g.state()
    g.salu(f"s_cmp_lt_i32 s{S_CNT}, 1")
    g.emit("s_cbranch_scc1 .Llast_%=", "branch")
    g.label(".Lloop_%=")
    loop_begin = len(g.out)
    iteration(g, "loop")
    g.salu(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    g.salu(f"s_cmp_lg_u32 s{S_CNT}, 0")
    g.emit("s_cbranch_scc1 .Lloop_%=", "branch")
    loop_end = len(g.out)
    if g.state() != st_in:
        raise RuntimeError(f"loop back-edge changes the outstanding-load queues:\n in  {st_in}\n out {g.state()}")
    g.label(".Llast_%=")
    last_begin = len(g.out)
    g.nop(11)
    iteration(g, "last")
    iteration(g, "drain")
    g.wait_lgkm(0)
    g.emit("s_barrier", "barrier")                           # the dQ tiles written below lie in ring slots other waves may still be reading
    epilogue(g)
    main = g.out
    check_hazards(main[:loop_end] + main[loop_begin:loop_end] + main[last_begin:])
    check_hazards(main[:first_end] + main[last_begin:])
    return g, dict(first=first_end, loop=loop_end - loop_begin, last=len(g.out) - last_begin)


def clobbers():
    c = [f"v{i}" for i in range(NV)] + [f"a{i}" for i in range(NA)] + [f"s{i}" for i in SGPRS] + ["vcc", "scc", "memory"]
    return ", ".join(f'"{x}"' for x in c)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "attn_dq_asm.inc")
    parts = ["// GENERATED by gen_dq_asm.py -- do not edit; edit the generator and re-run it (transception_amd.build does).\n"]
    for half in ("bf16", "f16"):
        g, stats = generate(half)
        nm = sum(1 for i in g.out if i.kind == "mfma")
        parts.append(f"// {half}: {len(g.out)} lines, {nm} MFMAs; sections {stats}\n")
        parts.append(f"#define TC_ATTN_DQ_ASM_{half.upper()} R\"ASM(\n" + "\n".join(i.text for i in g.out) + "\n)ASM\"\n")
    parts.append(f"#define TC_ATTN_DQ_ASM_CLOBBERS {clobbers()}\n")
    text = "".join(parts)
    if not os.path.exists(out) or open(out).read() != text:
        with open(out, "w") as f:
            f.write(text)
    if "-v" in sys.argv:
        print(text)


if __name__ == "__main__":
    main()
