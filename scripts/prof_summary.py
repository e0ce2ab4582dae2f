"""Summarise a rocprofv3 kernel_stats.csv: per-family GPU time per step."""
import csv, sys
path, steps = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
tot = sum(int(r['TotalDurationNs']) for r in rows)
ncalls = sum(int(r['Calls']) for r in rows)
print(f"total GPU kernel time {tot/1e6:.1f} ms over {steps} steps = {tot/steps/1e6:.2f} ms/step, {ncalls/steps:.0f} launches/step")
fam = {}
def key(n):
    for k, pat in (('gemm_bf16','gemm_bf16_kernel'),('gemm_f32','gemm_kernel'),('dw_strip','dw_strip'),('dw_wgrad(s2)','dw_wgrad'),('dw(s2)','dw_kernel'),
                   ('attn','attn_'),('ln','ln_'),('bn','bn_'),('softmax','softmax'),('colsum','colsum')):
        if pat in n: return k
    if 'at::' in n or 'rocclr' in n or 'Cijk' in n: return 'torch/runtime'
    return 'misc'
for r in rows:
    k = key(r['Name']); fam.setdefault(k, [0, 0]); fam[k][0] += int(r['TotalDurationNs']); fam[k][1] += int(r['Calls'])
for k, (v, c) in sorted(fam.items(), key=lambda x: -x[1][0]):
    print(f"  {k:16s} {v/steps/1e6:8.2f} ms/step {100*v/tot:5.1f}%  {c/steps:7.0f} calls/step  avg {v/c/1e3:7.1f} us")
print("top kernels:")
for r in sorted(rows, key=lambda r: -int(r['TotalDurationNs']))[:12]:
    print(f"  {int(r['TotalDurationNs'])/steps/1e6:7.2f} ms/step  {int(r['Calls'])/steps:6.0f}x  avg {float(r['AverageNs'])/1e3:7.1f} us  {r['Name'][:110]}")
