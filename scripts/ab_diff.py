"""Per-kernel difference of two rocprofv3 kernel_stats.csv files (scripts/ab_profile.sh): python scripts/ab_diff.py old.csv new.csv"""
import csv, re, sys
def load(p): return {r['Name']: (int(r['TotalDurationNs']), int(r['Calls'])) for r in csv.DictReader(open(p))}
a, b = load(sys.argv[1]), load(sys.argv[2])
rows = [(b.get(k, (0, 0))[0] - a.get(k, (0, 0))[0], k, a.get(k, (0, 0)), b.get(k, (0, 0))) for k in set(a) | set(b)]
for d, k, ta, tb in sorted(rows, key=lambda r: -abs(r[0]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 14]:
    name = re.sub(r'\s+', ' ', k)[:110]
    print(f"{d/1e6:+8.2f} ms  old {ta[0]/1e6:8.2f}/{ta[1]}  new {tb[0]/1e6:8.2f}/{tb[1]}  {name}")
print("total", sum(v[0] for v in a.values()) / 1e6, sum(v[0] for v in b.values()) / 1e6)
