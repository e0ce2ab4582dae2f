"""Per-section timeline of one replayed training step.

    TC_SEG_MARKS=gpurun_out/seg/labels.json rocprofv3 --kernel-trace --output-format csv -d gpurun_out/seg/trace -o t -- \
        python bench.py --no-cpu --no-side --steps 3 --warmup 1
    python scripts/seg_timeline.py gpurun_out/seg/trace gpurun_out/seg/labels.json [--json out.json] [--top N]

engine.Graph.segment() puts an empty marker launch (seg_marker_kernel, grid size = id + 1) at every section boundary of the forward
sweep and -- through the tape -- of the backward sweep; this script cuts the kernel trace of the last complete step at those markers
and prints launches / kernel time per section (forward and backward), with the top kernels of each.
"""
import collections
import csv
import glob
import json
import re
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("unsigned short", "bf16")
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\([^()]*\)$", "", n)
    return n[:70]


def main():
    trace_dir, labels_path = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 5
    f = sorted(glob.glob(trace_dir + "/**/*kernel_trace.csv", recursive=True))[0]
    labels = json.load(open(labels_path))
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    ev = []
    for r in rows:
        name = r["Kernel_Name"]
        gid = None
        if "seg_marker_kernel" in name:
            wg = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 64)) or 64)
            gid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0))) // max(wg, 1) - 1
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(name), gid))
    zeros = [i for i, e in enumerate(ev) if e[3] == 0]
    assert len(zeros) >= 2, "need two complete replayed steps in the trace"
    # the last complete step: from one id-0 marker back to the kernels that precede it (input copies etc.) is not attributed;
    # a step = [marker 0, next marker 0)
    a, b = zeros[-2], zeros[-1]
    seg = ev[a:b]
    sections = collections.OrderedDict()
    cur = None
    for s, e, n, gid in seg:
        if gid is not None:
            cur = labels[gid] if gid < len(labels) else f"?{gid}"
            sections.setdefault(cur, {"ms": 0.0, "launches": 0, "k": collections.defaultdict(lambda: [0.0, 0])})
            continue
        sec = sections[cur]
        sec["ms"] += (e - s) / 1e6
        sec["launches"] += 1
        sec["k"][n][0] += (e - s) / 1e6
        sec["k"][n][1] += 1
    wall = (seg[-1][1] - seg[0][0]) / 1e6
    tot = sum(v["ms"] for v in sections.values())
    nl = sum(v["launches"] for v in sections.values())
    print(f"step wall {wall:.3f} ms (includes {len(sections)} marker launches), kernel time {tot:.3f} ms, {nl} launches")
    out = {"wall_ms_with_markers": wall, "kernel_ms": tot, "launches": nl, "sections": {}}
    # pair F:x and B:x
    names = []
    for k in sections:
        nm = k[2:]
        if nm not in names:
            names.append(nm)
    print(f"{'section':10s} {'fwd ms':>8s} {'n':>4s} {'bwd ms':>8s} {'n':>4s} {'total':>8s} {'%':>5s}")
    for nm in names:
        fw, bw = sections.get("F:" + nm), sections.get("B:" + nm)
        fm, fn = (fw["ms"], fw["launches"]) if fw else (0.0, 0)
        bm, bn = (bw["ms"], bw["launches"]) if bw else (0.0, 0)
        print(f"{nm:10s} {fm:8.3f} {fn:4d} {bm:8.3f} {bn:4d} {fm + bm:8.3f} {100 * (fm + bm) / tot:5.1f}")
        out["sections"][nm] = {"fwd_ms": fm, "fwd_launches": fn, "bwd_ms": bm, "bwd_launches": bn}
    for k, v in sections.items():
        print(f"--- {k}: {v['ms']:.3f} ms, {v['launches']} launches")
        for n, (ms, c) in sorted(v["k"].items(), key=lambda x: -x[1][0])[:top]:
            print(f"      {ms * 1e3:8.1f} us {c:4d}x avg {ms * 1e3 / c:6.1f}  {n}")
    if "--json" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
