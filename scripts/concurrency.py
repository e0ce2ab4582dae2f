"""How full is the GPU during one replayed step?  From a rocprofv3 kernel trace: wall time with 0 / 1 / 2 / 3+ kernels in
flight, per-queue busy time, and the time-weighted number of workgroups resident (a kernel with < 256 workgroups leaves CUs idle)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'stem_im2col' in r['Kernel_Name']]
seg = rows[starts[-2]:starts[-1]]
t0 = int(seg[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in seg)
ev = []
for r in seg:
    wg = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])) * (int(r['Grid_Size_Y']) // int(r['Workgroup_Size_Y'])) * (int(r['Grid_Size_Z']) // int(r['Workgroup_Size_Z']))
    ev.append((int(r['Start_Timestamp']), 1, wg)); ev.append((int(r['End_Timestamp']), -1, -wg))
ev.sort()
depth = 0; wgs = 0; last = t0
hist = collections.Counter(); small = 0
for t, d, w in ev:
    dt = t - last
    hist[min(depth, 4)] += dt
    if depth > 0 and wgs < 256: small += dt
    depth += d; wgs += w; last = t
print(f"wall {(t1-t0)/1e6:.2f} ms")
for k in sorted(hist): print(f"  {k}{'+' if k == 4 else ' '} kernels in flight: {hist[k]/1e6:6.2f} ms")
print(f"  time with < 256 workgroups in flight (and >= 1 kernel): {small/1e6:.2f} ms")
q = collections.defaultdict(int)
for r in seg: q[r['Queue_Id']] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for k, v in sorted(q.items(), key=lambda x: -x[1]): print(f"  queue {k}: busy {v/1e6:.2f} ms")
