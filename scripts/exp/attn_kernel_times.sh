#!/bin/bash
# per-kernel averages of scripts/bench_attn.py under rocprofv3 for each library given (TC_LIB_PATH); "" = the in-tree build
cd /tmp; export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf /tmp/pp_k; TC_LIB_PATH=${lib:+/root/repo/scripts/exp/$lib} timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pp_k --output-format csv -- python /root/repo/scripts/bench_attn.py >/dev/null 2>&1
  python - "$lib" <<PY
import csv,glob,sys
f=glob.glob("/tmp/pp_k/**/*kernel_stats.csv",recursive=True)[0]
row={}
for r in csv.DictReader(open(f)):
    n=r["Name"]
    for k in ("dkv_asm","dkv_seg","dq_asm","dq_seg","fwd_asm","dkv_store"):
        if k in n: row[k]=float(r["AverageNs"])/1e3
print(sys.argv[1] or "in-tree", " ".join(f"{k}={v:.1f}" for k,v in sorted(row.items())))
PY
done
