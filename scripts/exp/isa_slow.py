"""Quarter-rate integer multiplies (v_mul_lo_u32 / v_mul_hi_u32 / v_mad_u64_u32 / v_mul_lo_i32) and division reciprocals per kernel of a hipcc -S
listing, with the share that sits in loop bodies (blocks at or after the target of a backward branch).  python scripts/exp/isa_slow.py x.s [regex]"""
import re, sys
rx = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
name, lines = None, []
def flush():
    if not name or (rx and not rx.search(name)) or "f16_t" in name:
        return
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            labels[m.group(1)] = i
    inloop = [False] * len(lines)
    for i, l in enumerate(lines):
        m = re.search(r"s_cbranch\w*\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)", l)
        if m:
            t = labels.get(m.group(1) or m.group(2))
            if t is not None and t <= i:
                for k in range(t, i + 1):
                    inloop[k] = True
    slow = ("v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mul_lo_i32", "v_mul_hi_i32", "v_mad_i64_i32")
    tot = sum(1 for l in lines if l.startswith("\t") and l.split() and not l.strip().startswith((".", ";")))
    n = sum(1 for l in lines if l.split() and l.split()[0] in slow)
    nl = sum(1 for i, l in enumerate(lines) if inloop[i] and l.split() and l.split()[0] in slow)
    r = sum(1 for l in lines if "v_rcp_iflag" in l)
    totl = sum(1 for i, l in enumerate(lines) if inloop[i] and l.startswith("\t") and l.split() and not l.strip().startswith((".", ";")))
    if n >= 8:
        print(f"{name[20:100]:80s} instrs {tot:5d} (in loops {totl:5d})  slow int mul {n:4d} (in loops {nl:4d})  rcp_iflag {r}")
for l in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        flush(); name, lines = m.group(1), []
    elif name:
        lines.append(l.rstrip("\n"))
        if "s_endpgm" in l:
            flush(); name = None
