cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python /root/repo/bench.py --no-side --no-cpu --steps 60 --warmup 10 > /dev/null 2>&1
python - <<'PY'
import csv, glob
f=sorted(glob.glob("/tmp/ks/**/*kernel_stats.csv", recursive=True))[-1]
rows=[(float(r["TotalDurationNs"]), int(r["Calls"]), float(r["AverageNs"])/1e3, r["Name"]) for r in csv.DictReader(open(f))]
tot=sum(r[0] for r in rows)
for t,c,a,n in sorted(rows, reverse=True)[:36]: print(f"{100*t/tot:5.2f}% {c:6d} x {a:7.1f} us  {n[:86]}")
print("---")
for t,c,a,n in rows:
    if any(k in n for k in ("fold","bn_","reduce")): print(f"{100*t/tot:5.2f}% {c:6d} x {a:7.1f} us  {n[:86]}")
PY
