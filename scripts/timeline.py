"""Section timeline of one training step from a rocprofv3 kernel trace (graph replay): milestones by marker kernels."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
def short(n):
    m = re.search(r'(gemm_bf16_kernel|gemm_kernel|dw_strip_kernel<[^>]*>|dw_\w+|ln_\w+|softmax_\w+|attn_\w+|delta_\w+|bn_\w+|\w+_kernel)', n)
    return m.group(1) if m else n[:40]
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows]
starts = [i for i, e in enumerate(ev) if e[2] == 'stem_im2col_kernel']
a, b = starts[-2], starts[-1]                       # one full replayed step
seg = ev[a:b]
t0 = seg[0][0]
print(f"step wall {(seg[-1][1]-t0)/1e6:.2f} ms, {len(seg)} launches, sum kernel {sum(e-s for s,e,_ in seg)/1e6:.2f} ms")
marks = ['stem_im2col_kernel', 'bn_partial_kernel', 'coord_pool_fwd_kernel', 'transpose_kernel', 'attn_fwd_seg_kernel', 'pixel_shuffle_kernel',
         'seg_loss_fwd_kernel', 'seg_loss_bwd_kernel', 'attn_bwd_dq_seg_kernel', 'coord_gate_bwd_att_kernel', 'sgd_kernel', 'patchify_kernel']
last = {}
for s, e, n in seg:
    if n in marks:
        k = last.get(n, 0); last[n] = k + 1
        print(f"  t={(s-t0)/1e6:7.2f} ms  {n} #{k}")

# per-kernel-family time inside windows
import collections
def window(lo, hi, label):
    c = collections.defaultdict(lambda: [0, 0])
    for s, e, n in seg:
        t = (s - t0) / 1e6
        if lo <= t < hi:
            c[n][0] += e - s; c[n][1] += 1
    tot = sum(v[0] for v in c.values())
    print(f"--- {label}: {tot/1e6:.2f} ms kernel time in window [{lo},{hi}) ms")
    for n, (v, k) in sorted(c.items(), key=lambda x: -x[1][0])[:14]:
        print(f"    {v/1e6:6.2f} ms  {k:4d}x  avg {v/k/1e3:7.1f} us  {n}")
if len(sys.argv) > 2:
    for w in sys.argv[2:]:
        lo, hi = map(float, w.split(':'))
        window(lo, hi, w)
