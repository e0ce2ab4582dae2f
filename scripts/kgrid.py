import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
pat=sys.argv[2:]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if any(p in n for p in pat):
        d[(n[:70],r["Grid_Size_X"],r["Workgroup_Size_X"])].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(d.items()):
    print(k, len(v), "avg %.2f us"%(sum(v)/len(v)/1e3), "tot %.1f"%(sum(v)/1e3))
