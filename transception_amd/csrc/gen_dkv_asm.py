#!/usr/bin/env python3
"""Emits the hand-scheduled gfx950 instruction stream of the bridge SR-attention dK / dV kernel (attn_bwd_dkv_asm_kernel in
attention_seg.hip), with the macro assembler of gen_attn_asm.py (counted waits, wait-state padding, back-edge queue check).

The compiler-scheduled kernel (attn_bwd_dkv_seg_kernel) holds 210 registers, so two waves share a SIMD and neither covers the other's
dependent chain S/dP MFMAs -> exp2 / multiply / pack -> dV^T/dK^T MFMAs: 78 us per launch at 224^2 B=16, a third of what its
instruction count allows (source-level software pipelining, bigger stages and accumulator initialisation were all measured equal --
DESIGN.md section 5).  This stream is the forward's structure with the roles swapped: the wave OWNS 32 keys (K / V fragments and the
dK^T / dV^T accumulators stay in AGPRs) and the 32-query sub-tiles of Q | dO | (-lse log2 e) | (-delta) stream through an LDS ring
shared by the four waves of the workgroup:

  iteration j:  ring store of sub-tile j+AHEAD-1 (requested one iteration ago), global request of sub-tile j+AHEAD
                dV^T(j), dK^T(j)   8 MFMAs, carrying exp2 / multiply of sub-tile j+1 and the row-fragment + statistics reads of sub-tile j+2
                pack P(j+1), dS(j+1)
                S(j+2), dP(j+2)    8 MFMAs (C operands = the statistics: P = exp2(S), dS = P dP with nothing else per score), carrying the
                                   transpose reads of sub-tile j+1 and the bookkeeping

One pool of eight 4-register fragments serves both fragment kinds: MFMA i of a group reads fragment i and the load that follows it
refills the same registers with the fragment MFMA i of the NEXT group needs.

Arithmetic = attn_bwd_dkv_seg_kernel<H, 4, true> (Q stored as q * scale * log2 e): dV = P^T dO, dK = ln 2 * dS^T Q, dS = P (dP - delta).

    python gen_dkv_asm.py            # writes attn_dkv_asm.inc next to this file
"""
from __future__ import annotations

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_attn_asm import Gen, Ins, areg, check_hazards, cvt_name, regs, vreg  # noqa: E402

# layout shared with attn_bwd_dkv_asm_kernel (keep in sync with the DA_* constants there) --------------------------------------------
LDR_B = 144                      # bytes per LDS row: 64 halfs + 8 pad
TILE_B = 32 * LDR_B               # one 32-row operand tile
SLOT_B = 2 * TILE_B + 256         # ring slot: Q tile | dO tile | 32 x (-lse log2 e) | 32 x (-delta)
OFF_G, OFF_L, OFF_D = TILE_B, 2 * TILE_B, 2 * TILE_B + 128
ABL = os.environ.get("TC_DKV_ABLATE", "")    # timing experiments only (wrong results): novalu, nopack, nolds, nostage, nobar, noprio, nosalu
NSLOT, AHEAD, PERIOD = 8, 6, 2    # stores run AHEAD - 1 sub-tiles ahead: AHEAD - 1 >= PERIOD + 2, NSLOT >= AHEAD - 1 + PERIOD - 1 (gen_attn_asm.py)
RED_B = 64 * 33 * 4               # one [d][key] fp32 accumulator tile in the epilogue's LDS layout

# VGPRs
S, DP, NEGL, NEGD = 0, 16, 32, 48
PP, PD = 64, 72                   # packed P / dS: two B operands of 4 each
STQ, STG, STS = 80, 84, 88        # ring staging
T0, T1 = 92, 93
AR, AT, AS, AW, AWS = 94, 95, 96, 97, 98
XL, XD = 99, 100                  # the sub-tile's 32 + 32 statistics, one per lane (lanes 16..31 / 48..63 repeat their row of 16)
PRM = S                           # 16 parameter dwords (prologue only: aliases the score registers)
NV = 101
# AGPRs
def DK(blk): return 16 * blk
def DV(blk): return 32 + 16 * blk
def KF(ks): return 64 + 4 * ks
def VF(ks): return 80 + 4 * ks
def F(i): return 96 + 4 * i
NA = 128
# operands
(OP_VQ, OP_VG, OP_VS, OP_WQ, OP_WS, OP_RB, OP_TB, OP_SB, OP_KO, OP_VO, OP_RED, OP_PRM,
 OP_RQ, OP_RG, OP_RS, OP_RK, OP_RV, OP_SMASK) = (f"%{i}" for i in range(18))
# SGPRs (parameters first, in the order of the parameter block)
(S_NSUB, S_T, S_OS, S_T1, S_T2, S_T3, S_A0, S_A1, S_A2, S_A3, S_LDQ, S_LDG, S_KSC) = range(36, 49)
S_CNT, S_PH, S_SR, S_ST, S_SW, S_A, S_X, S_OQ, S_OG = range(49, 58)
SGPRS = list(range(36, 58))
NPRM = 13


def mf_sdp(g, i):
    """MFMA i of the S / dP group: i = 2 ks + which; fragment i holds the Q (which = 0) / dO (1) rows, d slice ks."""
    ks, which = i >> 1, i & 1
    d, c0, b = (S, NEGL, KF(ks)) if which == 0 else (DP, NEGD, VF(ks))
    g.mfma(d, F(i), b, c0 if ks == 0 else d, b_acc=True)


def row_load(g, i, base):
    ks, which = i >> 1, i & 1
    g.emit(f"ds_read_b128 {areg(F(i), 4)}, {vreg(base)} offset:{(OFF_G if which else 0) + 32 * ks}", "ds_read", [vreg(base)], regs("a", F(i), 4))


def mf_dvdk(g, i):
    """MFMA i of the dV^T / dK^T group: i = 4 k2 + 2 which + blk; fragment i holds dO^T (which = 0) / Q^T (1), queries k2, d block blk."""
    k2, which, blk = i >> 2, (i >> 1) & 1, i & 1
    acc = DV(blk) if which == 0 else DK(blk)
    g.mfma(acc, F(i), (PP if which == 0 else PD) + 4 * k2, acc, acc_d=True)


def tr_load(g, i, base):
    k2, which, blk = i >> 2, (i >> 1) & 1, i & 1
    off = (0 if which else OFF_G) + 64 * blk
    for e in range(2):
        g.emit(f"ds_read_b64_tr_b16 {areg(F(i) + 2 * e, 2)}, {vreg(base)} offset:{off + (2 * k2 + e) * LDR_B}", "ds_read",
               [vreg(base)], regs("a", F(i) + 2 * e, 2))


def stat_loads(g, base):
    """The C operands of the S / dP MFMAs: register r of a lane is the statistic of query 16 h + r, the same for the 32 lanes of a half.
    ONE 4-byte LDS read per lane and matrix (lane (i, h) reads query 16 h + (i & 15)) and 16 row broadcasts (DPP row_newbcast) build each
    tuple -- 0.5 KB of LDS traffic per wave and sub-tile instead of the 8 KB of eight 16-byte reads (the stream is LDS-bandwidth-bound:
    removing every MFMA and VALU instruction left 44 of its 59 us)."""
    b = base if isinstance(base, str) else vreg(base)
    if "b128stats" in ABL:        # the first version: eight 16-byte reads (needs the statistics base at + 64 h instead of + 4 (16 h + lane & 15))
        ld = []
        for q in range(4):
            ld.append(lambda q=q: g.emit(f"ds_read_b128 {vreg(NEGL + 4 * q, 4)}, {b} offset:{16 * q}", "ds_read", [b], regs("v", NEGL + 4 * q, 4)))
            ld.append(lambda q=q: g.emit(f"ds_read_b128 {vreg(NEGD + 4 * q, 4)}, {b} offset:{128 + 16 * q}", "ds_read", [b], regs("v", NEGD + 4 * q, 4)))
        return ld, []
    ld = [lambda: g.emit(f"ds_read_b32 {vreg(XL)}, {b}", "ds_read", [b], [vreg(XL)]),
          lambda: g.emit(f"ds_read_b32 {vreg(XD)}, {b} offset:128", "ds_read", [b], [vreg(XD)])]
    mv = []
    for r in range(16):
        mv.append(lambda r=r: g.emit(f"v_mov_b32_dpp {vreg(NEGL + r)}, {vreg(XL)} row_newbcast:{r} row_mask:0xf bank_mask:0xf", "permlane", [vreg(XL)], [vreg(NEGL + r)]))
    for r in range(16):
        mv.append(lambda r=r: g.emit(f"v_mov_b32_dpp {vreg(NEGD + r)}, {vreg(XD)} row_newbcast:{r} row_mask:0xf bank_mask:0xf", "permlane", [vreg(XD)], [vreg(NEGD + r)]))
    return ld, mv


def valu_items(g):
    """P = exp2(S) in place, dS = P dP in place (the multiplies trail the exps: dP's last MFMA is the youngest)."""
    it = [lambda r=r: g.valu(f"v_exp_f32_e32 {vreg(S + r)}, {vreg(S + r)}", [vreg(S + r)], [vreg(S + r)], trans=True) for r in range(16)]
    it += [lambda r=r: g.valu(f"v_mul_f32_e32 {vreg(DP + r)}, {vreg(DP + r)}, {vreg(S + r)}", [vreg(DP + r), vreg(S + r)], [vreg(DP + r)]) for r in range(16)]
    return it


def pack_items(g):
    it = [lambda q=q: g.valu(f"{cvt_name(g.half)} {vreg(PP + q)}, {vreg(S + 2 * q)}, {vreg(S + 2 * q + 1)}", regs("v", S + 2 * q, 2), [vreg(PP + q)]) for q in range(8)]
    it += [lambda q=q: g.valu(f"{cvt_name(g.half)} {vreg(PD + q)}, {vreg(DP + 2 * q)}, {vreg(DP + 2 * q + 1)}", regs("v", DP + 2 * q, 2), [vreg(PD + q)]) for q in range(8)]
    return it


def interleave(g, n, mf, after, free, head=()):
    """head; then for i < n: mf(i), after(i), and an even share of the `free` closures."""
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


def stage_request(g):
    """Global loads of sub-tile j + AHEAD (tile S_T of the image): first row = A_s + 32 t with s the segment of t."""
    g.salu(f"s_cmp_ge_u32 s{S_T}, s{S_T1}")
    g.salu(f"s_cselect_b32 s{S_A}, s{S_A1}, s{S_A0}")
    g.salu(f"s_cmp_ge_u32 s{S_T}, s{S_T2}")
    g.salu(f"s_cselect_b32 s{S_A}, s{S_A2}, s{S_A}")
    g.salu(f"s_cmp_ge_u32 s{S_T}, s{S_T3}")
    g.salu(f"s_cselect_b32 s{S_A}, s{S_A3}, s{S_A}")
    g.salu(f"s_lshl_b32 s{S_X}, s{S_T}, 5")
    g.salu(f"s_add_u32 s{S_A}, s{S_A}, s{S_X}")
    g.salu(f"s_mul_i32 s{S_OQ}, s{S_A}, s{S_LDQ}")
    g.salu(f"s_mul_i32 s{S_OG}, s{S_A}, s{S_LDG}")
    g.valu(f"v_add_u32_e32 {vreg(T0)}, s{S_OQ}, {OP_VQ}", [], [vreg(T0)])
    g.emit(f"buffer_load_dwordx4 {vreg(STQ, 4)}, {vreg(T0)}, {OP_RQ}, 0 offen", "vmem_load", [vreg(T0)], regs("v", STQ, 4))
    g.valu(f"v_add_u32_e32 {vreg(T1)}, s{S_OG}, {OP_VG}", [], [vreg(T1)])
    g.emit(f"buffer_load_dwordx4 {vreg(STG, 4)}, {vreg(T1)}, {OP_RG}, 0 offen", "vmem_load", [vreg(T1)], regs("v", STG, 4))
    g.valu(f"v_add_u32_e32 {vreg(T0)}, s{S_OS}, {OP_VS}", [], [vreg(T0)])
    g.salu(f"s_mov_b64 exec, {OP_SMASK}")
    g.emit(f"buffer_load_dwordx4 {vreg(STS, 4)}, {vreg(T0)}, {OP_RS}, 0 offen", "vmem_load", [vreg(T0)], regs("v", STS, 4))
    g.salu("s_mov_b64 exec, -1")
    g.salu(f"s_add_u32 s{S_T}, s{S_T}, 1")
    g.salu(f"s_add_u32 s{S_OS}, s{S_OS}, 256")


def stash(g):
    g.wait_vm(0)
    g.valu(f"v_add_u32_e32 {vreg(AW)}, s{S_SW}, {OP_WQ}", [], [vreg(AW)])
    g.valu(f"v_add_u32_e32 {vreg(AWS)}, s{S_SW}, {OP_WS}", [], [vreg(AWS)])
    g.emit(f"ds_write_b128 {vreg(AW)}, {vreg(STQ, 4)}", "ds_write", [vreg(AW)] + regs("v", STQ, 4))
    g.emit(f"ds_write_b128 {vreg(AW)}, {vreg(STG, 4)} offset:{OFF_G}", "ds_write", [vreg(AW)] + regs("v", STG, 4))
    g.salu(f"s_mov_b64 exec, {OP_SMASK}")
    g.emit(f"ds_write_b128 {vreg(AWS)}, {vreg(STS, 4)}", "ds_write", [vreg(AWS)] + regs("v", STS, 4))
    g.salu("s_mov_b64 exec, -1")


def slots(g, which):
    for s in which:
        g.salu(f"s_add_u32 s{s}, s{s}, {SLOT_B}")
        g.salu(f"s_cmp_eq_u32 s{s}, {SLOT_B * NSLOT}")
        g.salu(f"s_cselect_b32 s{s}, 0, s{s}")


def iteration(g, mode):
    """mode 'first' (j = -1): no dV / dK MFMAs; 'loop'; 'last' (j = nsub - 2): no sub-tile j + 2; 'drain' (j = nsub - 1): dV / dK only."""
    first, last, drain = mode == "first", mode == "last", mode == "drain"
    stage = not (last or drain)
    head = []
    if mode == "loop":
        def bar():
            g.wait_lgkm(0)
            if "nobar" not in ABL:
                g.salu(f"s_cmp_lg_u32 s{S_PH}, 0")
                g.emit(f"s_cbranch_scc1 .Lnobar{g.uid}_%=", "branch")
                g.emit("s_barrier", "barrier")
                g.label(f".Lnobar{g.uid}_%=")
                g.uid += 1
            g.salu(f"s_add_u32 s{S_PH}, s{S_PH}, 1")
            g.salu(f"s_and_b32 s{S_PH}, s{S_PH}, {PERIOD - 1}")
        head.append(bar)
    if stage and not ("nostage" in ABL and mode == "loop"):
        if not first:
            head.append(lambda: stash(g))                  # the sub-tile requested one iteration ago: a whole iteration of latency cover
        head.append(lambda: stage_request(g))
    if stage:
        head.append(lambda: g.valu(f"v_add_u32_e32 {vreg(AR)}, s{S_SR}, {OP_RB}", [], [vreg(AR)]))
        head.append(lambda: g.valu(f"v_add_u32_e32 {vreg(AS)}, s{S_SR}, {OP_SB}", [], [vreg(AS)]))
    free = [] if (drain or "novalu" in ABL) else valu_items(g)
    nolds = "nolds" in ABL and mode == "loop"
    if stage and not nolds:
        ld, mv = stat_loads(g, AS)
        free = ld + free + mv
    if not first and "noprio" not in ABL:
        head.append(lambda: g.salu("s_setprio 1"))
    interleave(g, 8, None if first else (lambda i: mf_dvdk(g, i)), (lambda i: row_load(g, i, AR)) if (stage and not nolds) else (lambda i: None), free, head)
    if not first and "noprio" not in ABL:
        g.salu("s_setprio 0")
    if drain:
        return
    if "nopack" not in ABL:
        for f in pack_items(g):
            f()
    # ---- S / dP (j+2) group: transpose reads of sub-tile j+1 behind the MFMAs
    head = [lambda: g.valu(f"v_add_u32_e32 {vreg(AT)}, s{S_ST}, {OP_TB}", [], [vreg(AT)])]
    noop = lambda: None
    free = [noop] * 7 + [(lambda: slots(g, (S_SR, S_ST, S_SW))) if not last else noop]
    interleave(g, 8, None if last else (lambda i: mf_sdp(g, i)), (lambda i: None) if nolds else (lambda i: tr_load(g, i, AT)), free, head)


def prologue(g):
    # parameter block: 16 dwords in LDS -> SGPRs
    for q in range(4):
        g.emit(f"ds_read_b128 {vreg(PRM + 4 * q, 4)}, {OP_PRM} offset:{16 * q}", "ds_read", [], regs("v", PRM + 4 * q, 4))
    g.wait_lgkm(0)
    for k in range(NPRM):
        g.valu(f"v_readfirstlane_b32 s{S_NSUB + k}, {vreg(PRM + k)}", [vreg(PRM + k)], [])
    g.nop(5)                                                                            # VALU-written SGPRs before their first SALU / VMEM use
    # K / V fragments of the wave's 32 keys, straight from global memory into AGPRs
    for ks in range(4):
        g.emit(f"buffer_load_dwordx4 {areg(KF(ks), 4)}, {OP_KO}, {OP_RK}, 0 offen offset:{32 * ks}", "vmem_load", [], regs("a", KF(ks), 4))
        g.emit(f"buffer_load_dwordx4 {areg(VF(ks), 4)}, {OP_VO}, {OP_RV}, 0 offen offset:{32 * ks}", "vmem_load", [], regs("a", VF(ks), 4))
    g.salu(f"s_sub_u32 s{S_CNT}, s{S_NSUB}, 2")                                         # steady iterations j = 0 .. nsub - 3
    g.salu(f"s_mov_b32 s{S_PH}, {2 % PERIOD}")
    g.salu(f"s_mov_b32 s{S_SR}, {SLOT_B}")                                              # row fragments / statistics of sub-tile 1
    g.salu(f"s_mov_b32 s{S_ST}, 0")                                                     # transpose reads of sub-tile 0
    g.salu(f"s_mov_b32 s{S_SW}, {(AHEAD - 2) * SLOT_B}")                                # iteration 0 stores sub-tile AHEAD - 1 (requested by iteration -1)
    for r in range(64):
        g.valu(f"v_accvgpr_write_b32 {areg(r)}, 0", [], [areg(r)])
    # sub-tile 0: statistics, row fragments, S / dP
    ld, mv = stat_loads(g, OP_SB)
    for f in ld:
        f()
    for i in range(8):
        ks, which = i >> 1, i & 1
        g.emit(f"ds_read_b128 {areg(F(i), 4)}, {OP_RB} offset:{(OFF_G if which else 0) + 32 * ks}", "ds_read", [], regs("a", F(i), 4))
    for f in mv:
        f()
    g.wait_vm(0)
    for i in range(8):
        mf_sdp(g, i)


def epilogue(g):
    g.nop(32)
    for which in range(2):                                   # dK (scaled), then dV: [d][key] fp32 tiles, pitch 33
        for blk in range(2):
            for r in range(16):
                a = (DK if which == 0 else DV)(blk) + r
                g.valu(f"v_accvgpr_read_b32 {vreg(S + r)}, {areg(a)}", [areg(a)], [vreg(S + r)])
            if which == 0:
                for r in range(16):
                    g.valu(f"v_mul_f32_e32 {vreg(S + r)}, s{S_KSC}, {vreg(S + r)}", [vreg(S + r)], [vreg(S + r)])
            for r in range(16):
                off = which * RED_B + (((r & 3) + 8 * (r >> 2) + 32 * blk) * 33) * 4
                g.emit(f"ds_write_b32 {OP_RED}, {vreg(S + r)} offset:{off}", "ds_write", [vreg(S + r)])
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
    if g.state() != st_in and not ABL:
        raise RuntimeError(f"loop back-edge changes the outstanding-load queues:\n in  {st_in}\n out {g.state()}")
    g.label(".Llast_%=")
    last_begin = len(g.out)
    g.nop(11)
    iteration(g, "last")
    iteration(g, "drain")
    g.wait_lgkm(0)
    g.emit("s_barrier", "barrier")                           # the accumulator tiles below overwrite ring slots other waves may still read
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
    out = os.path.join(here, "attn_dkv_asm.inc")
    parts = ["// GENERATED by gen_dkv_asm.py -- do not edit; edit the generator and re-run it (transception_amd.build does).\n"]
    for half in ("bf16", "f16"):
        g, stats = generate(half)
        nm = sum(1 for i in g.out if i.kind == "mfma")
        parts.append(f"// {half}: {len(g.out)} lines, {nm} MFMAs; sections {stats}\n")
        parts.append(f"#define TC_ATTN_DKV_ASM_{half.upper()} R\"ASM(\n" + "\n".join(i.text for i in g.out) + "\n)ASM\"\n")
    parts.append(f"#define TC_ATTN_DKV_ASM_CLOBBERS {clobbers()}\n")
    text = "".join(parts)
    if not os.path.exists(out) or open(out).read() != text:
        with open(out, "w") as f:
            f.write(text)
    if "-v" in sys.argv:
        print(text)


if __name__ == "__main__":
    main()
