"""Top kernels of a rocprofv3 --stats run: python scripts/kstats.py <dir> [n]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for r in list(csv.DictReader(open(f)))[:n]:
    name = re.sub(r'\(.*', '', re.sub(r'\(anonymous namespace\)::', '', r['Name']))[:72]
    print(f"  {name:72s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Percentage']}%")
