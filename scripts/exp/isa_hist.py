"""Instruction histogram per basic block of one kernel in a hipcc -S listing (static counts: which loops are VALU-heavy, how many
v_mov / v_pk_* the compiler spent).   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast --cuda-device-only -S -o x.s file.hip
    python scripts/exp/isa_hist.py x.s <kernel-name-regex> [min-block-size]"""
import collections, re, sys

txt = open(sys.argv[1]).read().split("\n")
rx = re.compile(sys.argv[2])
minb = int(sys.argv[3]) if len(sys.argv) > 3 else 40
i = 0
while i < len(txt):
    m = re.match(r"^(_Z\w+):", txt[i])
    if not (m and rx.search(m.group(1))):
        i += 1
        continue
    name = m.group(1)
    blocks, cur, label = [], collections.Counter(), "entry"
    i += 1
    while i < len(txt) and "s_endpgm" not in txt[i]:
        l = txt[i]
        lm = re.match(r"^(\.LBB\w+):", l)
        if lm:
            blocks.append((label, cur)); cur, label = collections.Counter(), lm.group(1)
        elif l.startswith("\t") and l.split() and not l.strip().startswith((".", ";")):
            cur[l.split()[0]] += 1
        i += 1
    blocks.append((label, cur))
    tot = collections.Counter()
    for _, c in blocks:
        tot.update(c)
    print(f"== {name[:100]}: {sum(tot.values())} instructions")
    def cls(c):
        g = collections.Counter()
        for op, n in c.items():
            k = ("mfma" if "mfma" in op else "pk" if op.startswith("v_pk_") else "mov" if op in ("v_mov_b32", "v_mov_b64", "v_accvgpr_write_b32", "v_accvgpr_read_b32") else
                 "trans" if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)", op) else "cvt" if op.startswith(("v_cvt", "v_perm", "v_lshl", "v_lshr", "v_and", "v_or", "v_bfe", "v_bfi")) else
                 "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else
                 "salu" if op.startswith("s_") else "other")
            g[k] += n
        return dict(g)
    print("  total", cls(tot))
    for lab, c in blocks:
        n = sum(c.values())
        if n >= minb:
            print(f"  {lab:14s} {n:5d}", cls(c))
    break
