"""GEMM time of one replayed step grouped by (template instance, grid) from a rocprofv3 kernel trace."""
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
pat = sys.argv[2] if len(sys.argv) > 2 else 'gemm'
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'stem_im2col' in r['Kernel_Name']]
seg = rows[starts[-2]:starts[-1]]
c = collections.defaultdict(lambda: [0, 0])
for r in seg:
    n = r['Kernel_Name']
    if pat not in n: continue
    m = re.search(r'<(.*)>', n)
    key = (m.group(1) if m else n[:30], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
    c[key][0] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); c[key][1] += 1
tot = sum(v[0] for v in c.values())
print(f"total {tot/1e6:.2f} ms")
for k, (v, n) in sorted(c.items(), key=lambda x: -x[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"  {v/1e6:6.3f} ms {n:4d}x avg {v/n/1e3:7.1f} us  {k}")
