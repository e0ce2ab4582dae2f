"""Longest launches of one replayed training step from a rocprofv3 kernel trace: python scripts/slow_launches.py <dir> [min_us]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'stem_im2col_kernel' in r['Kernel_Name']]
seg = rows[starts[-2]:starts[-1]]
t0 = int(seg[0]['Start_Timestamp'])
tot = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
print(f"{len(seg)} launches, {tot/1e6:.2f} ms of kernel time")
acc = 0
for r in seg:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if d >= thr:
        acc += d
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); n = re.sub(r'\(.*', '', n)[:60]
        print(f"t={(int(r['Start_Timestamp'])-t0)/1e6:7.2f} ms {d:7.1f} us grid {r.get('Grid_Size_X', r.get('Grid_Size','?')):>8} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size','?')):>4} lds {r.get('LDS_Block_Size', '?'):>6}  {n}")
print(f"launches >= {thr} us: {acc/1e3:.2f} ms")
