#!/bin/bash
# Copy one collection of scripts/profile_round6.sh (gpurun_out/r6_final/) into profiles/ under its committed names and stamp the commit.
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r6_final
cp $O/r6_attn_pmc.json $O/r6_attn_kernel_stats.csv $O/r6_hbm_by_kernel.json $O/r6_step_timeline.json $O/r6_seg_timeline.json $O/r6_seg_timeline.txt \
   $O/r6_ripm_iff_hbm.json $O/r6_attn_sustained_power.txt profiles/
cp $O/stats_kernel_stats.csv profiles/r6_default_bench_kernel_stats.csv
cp $O/bench.json profiles/r6_bench_line_under_rocprof.json
cp $O/stats_c4_kernel_stats.csv profiles/r6_config4_512_b8_bf16_kernel_stats.csv
cp $O/bench_c4_512_b8_bf16.json profiles/r6_config4_bench_line.json
cp $O/stats_c5_kernel_stats.csv profiles/r6_config5_384_b8_f16_kernel_stats.csv
cp $O/bench_c5_384_b8_f16.json profiles/r6_config5_bench_line.json
cp $O/bench_split.json profiles/r6_bench_split_step.json
[ -s $O/bench_plain.json ] && cp $O/bench_plain.json profiles/r6_bench_line.json
python scripts/provenance.py commit profiles/r6_attn_pmc.json profiles/r6_hbm_by_kernel.json profiles/r6_step_timeline.json profiles/r6_seg_timeline.json profiles/r6_ripm_iff_hbm.json | cut -c1-90
python - <<'PY'
import json, sys
sys.path.insert(0, 'scripts')
import provenance
d = json.load(open('profiles/r6_step_timeline.json'))
print("stamp matches sources:", d['provenance']['source_sha'] == provenance.source_digest(), "bench:", d['provenance']['bench_sha'] == provenance.bench_digest())
print("step timeline: wall", d['wall_ms'], "launches", d['launches'], {k: round(v['ms'], 3) for k, v in d['families'].items()})
h = json.load(open('profiles/r6_hbm_by_kernel.json')); print("hbm GB/step", h['step_hbm_GB'])
b = json.load(open('profiles/r6_bench_line.json'))
print("bench: value", round(b['value'], 1), "ms", round(b['ms_per_step'], 3), "resident", round(b['config']['resident_batch_images_per_sec'], 1), "launches", b['config']['launches_per_step'])
print("roofline", round(b['roofline']['frac'], 4), b['roofline']['avg_launch_us'], "cold", b['roofline'].get('single_cold_replay_us'))
for k in ('roofline_attn_bwd_graph_replay', 'roofline_ripm', 'roofline_iff', 'roofline_step_dominant'):
    r = b.get(k) or {}
    print(k, r.get('frac'), r.get('avg_launch_us') or r.get('us_per_pass'))
print("step", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in b['step'].items() if k in ('traffic_over_algorithmic', 'launch_floor_ms', 'achieved_tflops', 'launches')})
for c in ('4', '5'):
    x = json.load(open(f'profiles/r6_config{c}_bench_line.json')); print("config", c, round(x['value'], 1), round(x['ms_per_step'], 2), round(x['roofline']['frac'], 3), round(x['roofline_attn_bwd_graph_replay']['frac'], 3))
x = json.load(open('profiles/r6_bench_split_step.json')); print("split exposed", x['config'].get('allreduce_exposed_ms'))
r = json.load(open('profiles/r6_bench_line_under_rocprof.json')); print("under rocprof", round(r['value'], 1), round(r['ms_per_step'], 3), round(r['config']['resident_batch_images_per_sec'], 1))
PY
cat profiles/r6_attn_sustained_power.txt
