"""Where do the __amd_rocclr_copyBuffer dispatches sit inside one replayed step?  python scripts/copybuffer_where.py <rocprof dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows]
starts = [i for i, e in enumerate(ev) if 'stem_im2col_kernel' in e[2]]
a, b = starts[-2], starts[-1]
seg = ev[a:b]
t0 = seg[0][0]
print(f"step wall {(seg[-1][1]-t0)/1e6:.2f} ms, {len(seg)} launches, sum kernel {sum(e-s for s,e,_ in seg)/1e6:.2f} ms")
gaps = sum(max(0, seg[i+1][0] - seg[i][1]) for i in range(len(seg) - 1))
print(f"sum of gaps between consecutive kernels {gaps/1e6:.2f} ms")
cp = [(i, s, e) for i, (s, e, n) in enumerate(seg) if 'copyBuffer' in n]
print(len(cp), "copyBuffer dispatches, total", sum(e - s for _, s, e in cp) / 1e6, "ms")
for i, s, e in cp[:40]:
    prev = seg[i - 1][2][:60] if i else ''
    nxt = seg[i + 1][2][:60] if i + 1 < len(seg) else ''
    print(f"  t={(s-t0)/1e6:7.3f} ms dur {(e-s)/1e3:5.1f} us  after [{prev}]  before [{nxt}]")
# also outside: between the end of this step and the next
