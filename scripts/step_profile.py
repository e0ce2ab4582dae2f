"""Per-step kernel timeline summary from a rocprofv3 kernel trace (rocpd .db or kernel_trace.csv): isolates ONE replayed training step
(between two consecutive stem_im2col launches late in the run) and prints per-kernel calls / time / average, plus family totals.

    python scripts/step_profile.py <results.db | kernel_trace.csv> [--json out.json] [--all]
"""
import collections
import csv
import json
import re
import sqlite3
import sys


def load(path):
    if path.endswith(".db"):
        db = sqlite3.connect(path)
        rows = db.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start").fetchall()
        return [(n, s, e, (gx, gy, gz, wx)) for n, s, e, gx, gy, gz, wx in rows]
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    return [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
             (int(r.get("Grid_Size_X", 0) or 0), int(r.get("Grid_Size_Y", 0) or 0), int(r.get("Grid_Size_Z", 0) or 0), int(r.get("Workgroup_Size_X", 0) or 0))) for r in rows]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([\w:]+(<[^(]*>)?)", n)
    s = m.group(1) if m else n
    s = s.replace("unsigned short", "bf16")
    return s[:90]


FAMILIES = [("gemm", r"^gemm"), ("dw/ffn_mid", r"^(dw_|ffn_mid)"), ("layernorm", r"^ln_"), ("batchnorm", r"^bn_"), ("attention", r"^(attn_|delta_)"),
            ("factor_att", r"^factor_att"), ("softmax", r"^softmax"), ("torch", r"^at::|rocclr|Cijk"), ("other", r".")]


def main():
    path = sys.argv[1]
    ev = load(path)
    stems = [i for i, e in enumerate(ev) if "stem_im2col" in e[0]]
    if len(stems) < 2 or "--all" in sys.argv:
        seg = ev
        steps = max(len(stems), 1)
    else:
        a, b = stems[-2], stems[-1]
        seg, steps = ev[a:b], 1
    wall = (seg[-1][2] - seg[0][1]) / 1e6
    tot = sum(e - s for _, s, e, _ in seg) / 1e6
    per = collections.OrderedDict()
    for n, s, e, g in seg:
        k = short(n)
        c = per.setdefault(k, [0, 0])
        c[0] += 1; c[1] += e - s
    print(f"{'one replayed step' if steps == 1 else f'whole trace / {steps} steps'}: wall {wall / steps:.3f} ms, sum of kernel time {tot / steps:.3f} ms, {len(seg) / steps:.0f} launches")
    fam = collections.OrderedDict((f, [0, 0]) for f, _ in FAMILIES)
    for k, (c, t) in per.items():
        for f, pat in FAMILIES:
            if re.search(pat, k):
                fam[f][0] += c; fam[f][1] += t
                break
    print("families:")
    for f, (c, t) in fam.items():
        print(f"  {f:12s} {c / steps:7.1f} launches  {t / steps / 1e6:7.3f} ms")
    print("kernels:")
    for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"  {t / steps / 1e6:7.3f} ms  {c / steps:6.1f}x  avg {t / c / 1e3:7.1f} us  {k}")
    if "--json" in sys.argv:
        out = {"wall_ms": wall / steps, "kernel_ms": tot / steps, "launches": len(seg) / steps,
               "families": {f: {"launches": c / steps, "ms": t / steps / 1e6} for f, (c, t) in fam.items()},
               "kernels": {k: {"launches": c / steps, "ms": t / steps / 1e6, "avg_us": t / c / 1e3} for k, (c, t) in per.items()}}
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
