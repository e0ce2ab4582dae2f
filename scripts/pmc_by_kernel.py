"""Per-kernel averages of a rocprofv3 --pmc run: python scripts/pmc_by_kernel.py <counter_collection.csv> [name filter]"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in rows:
    n = r["Kernel_Name"]
    if flt not in n: continue
    m = re.search(r'(\w+<[^(]*>|\w+)\(', n.replace("(anonymous namespace)::", ""))
    k = m.group(1) if m else n[:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r.get("Dispatch_Id", len(cnt[k])))
for k, v in agg.items():
    c = max(len(cnt[k]), 1)
    print(f"{k} ({c} launches): " + "  ".join(f"{a}={b/c:,.0f}" for a, b in sorted(v.items())))
