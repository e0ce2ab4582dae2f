"""Fold scripts/pmc_stage.sh's passes: per stage (RIPM, IFF; forward + backward of all three encoder stages, B=16 bf16) the kernel time of one
pass, its launches, the memory-side bytes (2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md's gfx950 correction), GB/s against the 8 TB/s
peak, and the algorithmic bytes of SURVEY.md 8(d) beside them."""
import csv, glob, json, sys

src, out = sys.argv[1], sys.argv[2]
PASSES = 3 + 1 + 1 + 10              # bench_stage.py --iters 10: 3 eager warm-ups, 1 eager pass on the side stream (the captured pass is not executed), 1 + 10 replays
res = {"what": "RIPM (Patch_Embed_stage, MSTr.py:704-732) and IFF (CoordAtt, MSTr.py:1304-1348) alone: forward + backward of the three encoder stages of each at "
               "B=16, 224^2, bf16 (scripts/bench_stage.py under rocprofv3: --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE; scripts/pmc_stage.sh)",
       "peak_GBps": 8000.0}
for st in ("ripm", "iff"):
    line = json.loads(open(f"{src}/{st}.json").read().strip().splitlines()[-1])
    stats = list(csv.DictReader(open(glob.glob(f"{src}/{st}/trace/**/*kernel_stats.csv", recursive=True)[0])))
    skip = lambda n: "at::" in n or "elementwise_kernel" in n or n.startswith("__amd_rocclr") or "cast_kernel" in n     # model construction, not the stage
    own = [r for r in stats if not skip(r["Name"])]
    calls = sum(int(r["Calls"]) for r in own)
    ns = sum(float(r["TotalDurationNs"]) for r in own)
    tot = {}
    for which, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        rows = csv.DictReader(open(glob.glob(f"{src}/{st}/{which}/**/*counter_collection.csv", recursive=True)[0]))
        tot[which] = sum(float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == name and not skip(r["Kernel_Name"])) * 1024.0
    per = calls / PASSES
    by = (2.0 * tot["fetch"] + tot["write"]) / PASSES
    us = ns / 1e3 / PASSES
    res[st] = {"launches_per_pass": per, "kernel_us_per_pass": us, "graph_replay_us_per_pass": line["us_per_fwd_bwd"],
               "traffic_bytes_per_pass": by, "traffic_GBps": by / us / 1e3, "traffic_frac_of_peak": by / us / 1e3 / 8000.0,
               "algorithmic_bytes_per_pass": line["algorithmic_bytes"], "algorithmic_GBps": line["algorithmic_bytes"] / us / 1e3,
               "algorithmic_frac_of_peak": line["algorithmic_bytes"] / us / 1e3 / 8000.0,
               "kernels": sorted(({"name": r["Name"][:90], "calls_per_pass": int(r["Calls"]) / PASSES, "avg_us": float(r["AverageNs"]) / 1e3} for r in own),
                                 key=lambda k: -k["calls_per_pass"] * k["avg_us"])[:12]}
json.dump(res, open(out, "w"), indent=1)
