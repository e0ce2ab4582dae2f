"""Which SOURCE lines the slow instructions of a kernel come from: compile with line tables
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -munsafe-fp-atomics -gline-tables-only --cuda-device-only -S -o x.s file.hip
and count, per `.loc` line, the instructions matching a regex (default: quarter-rate integer multiplies).
    python scripts/exp/isa_lines.py x.s <kernel-regex> [instr-regex]"""
import collections, re, sys
krx = re.compile(sys.argv[2])
irx = re.compile(sys.argv[3] if len(sys.argv) > 3 else r"^v_(mul_lo_u32|mul_hi_u32|mad_u64_u32|mul_lo_i32|mul_hi_i32|mad_i64_i32|rcp_iflag)")
files, cur, on = {}, None, False
cnt = collections.Counter()
tot = 0
for l in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        on = bool(krx.search(m.group(1))) and "f16_t" not in m.group(1)
        if on:
            print("==", m.group(1)[:110])
        continue
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[m.group(1)] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    if not on:
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(m.group(1), m.group(1)), int(m.group(2)))
        continue
    if "s_endpgm" in l:
        on = False
        continue
    t = l.split()
    if l.startswith("\t") and t and irx.search(t[0]):
        cnt[cur] += 1
        tot += 1
print("total", tot)
for k, v in cnt.most_common(40):
    print(f"{v:5d}  {k[0]}:{k[1]}")
