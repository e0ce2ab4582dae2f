"""Wall time vs kernel time of the last few replayed steps in a rocprofv3 kernel trace: python scripts/step_gaps.py <dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows]
starts = [i for i, e in enumerate(ev) if 'stem_im2col_kernel' in e[2]]
segs = [(a, b) for a, b in zip(starts, starts[1:]) if b - a > 500]
for a, b in segs[-8:]:
    seg = ev[a:b]
    wall = (ev[b][0] - seg[0][0]) / 1e6
    busy = sum(e - s for s, e, _ in seg) / 1e6
    gaps = sorted(((seg[i + 1][0] - seg[i][1]) / 1e3, seg[i][2][:40], seg[i + 1][2][:40]) for i in range(len(seg) - 1))[-3:]
    names = {}
    for s, e, n in seg:
        if any(k in n for k in ('copyBuffer', 'slice_augment', 'spline', 'zoom_sample')): names[n[:30]] = names.get(n[:30], 0) + (e - s) / 1e3
    for g, n0, n1 in gaps: print(f"      gap {g:7.0f} us after [{n0}] before [{n1}]")
    print(f"step wall {wall:6.2f} ms  kernels {busy:6.2f} ms  launches {len(seg)}  largest gaps(us) {[round(g[0]) for g in gaps]}  extras(us) { {k: round(v) for k, v in names.items()} }")
