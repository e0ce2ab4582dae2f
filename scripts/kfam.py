"""Per-kernel totals of ONE replayed step from a rocprofv3 kernel trace: python scripts/kfam.py <trace dir> <steps traced> [name filter ...]"""
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
steps = int(sys.argv[2])
pat = sys.argv[3:]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"^void ", "", n)[:60]
    if not pat or any(p in n for p in pat):
        d[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = 0
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v) / 1e3 / steps:9.1f} us/step {len(v) / steps:7.1f}x  avg {sum(v) / len(v) / 1e3:7.2f}  {k}")
    tot += sum(v)
print(f"{tot / 1e3 / steps:9.1f} us/step total")
