"""Kernels of a rocprofv3 --stats csv whose name matches a regex: python scripts/kgrep.py <dir-or-csv> <regex>"""
import csv, glob, os, re, sys
f = sys.argv[1]
if os.path.isdir(f):
    f = (glob.glob(f + "/*kernel_stats.csv") + glob.glob(f + "/*/*kernel_stats.csv"))[0]
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::", "", r["Name"]))[:70]
    if re.search(sys.argv[2], name):
        print(f"{name:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs']) / 1e3:8.1f} us  min {float(r['MinNs']) / 1e3:8.1f}")
