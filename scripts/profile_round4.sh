# round-4 profile set (one gpurun call): attention counters (4 passes), per-kernel HBM traffic of one step (3 passes), kernel stats of
# the default bench command + the bench line, kernel stats of BASELINE configs 5 (384^2 B=8 fp16) and 4 (512^2 B=8 bf16)
cd /root/repo
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash scripts/pmc_attn.sh pmc_attn_r4 > gpurun_out/pmc_attn_r4.log 2>&1
python scripts/pmc_summary.py gpurun_out/pmc_attn_r4/attn_pmc.json gpurun_out/pmc_attn_r4/fetch gpurun_out/pmc_attn_r4/write gpurun_out/pmc_attn_r4/sq > gpurun_out/pmc_attn_r4/summary.txt 2>&1
cp $(find gpurun_out/pmc_attn_r4/trace -name "*kernel_stats.csv" | head -1) gpurun_out/pmc_attn_r4/attn_kernel_stats.csv
find gpurun_out/pmc_attn_r4 -name "*kernel_trace.csv" -delete; find gpurun_out/pmc_attn_r4 -name "*counter_collection.csv" -delete; find gpurun_out/pmc_attn_r4 -name "*agent_info.csv" -delete
bash scripts/pmc_step.sh pmc_step_r4 > gpurun_out/pmc_step_r4.log 2>&1
python scripts/hbm_by_kernel.py gpurun_out/pmc_step_r4 gpurun_out/pmc_step_r4/hbm_by_kernel.json > gpurun_out/pmc_step_r4/hbm.txt 2>&1
python scripts/step_profile.py $(find gpurun_out/pmc_step_r4/trace -name "*kernel_trace.csv" | head -1) --json gpurun_out/pmc_step_r4/step_timeline.json > gpurun_out/pmc_step_r4/step_profile.txt 2>&1
find gpurun_out/pmc_step_r4 -name "*kernel_trace.csv" -delete; find gpurun_out/pmc_step_r4 -name "*counter_collection.csv" -delete; find gpurun_out/pmc_step_r4 -name "*agent_info.csv" -delete
O=gpurun_out/r4_final; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats -- python /root/repo/bench.py > /root/repo/$O/bench.json 2> /root/repo/$O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats_c5 -- python /root/repo/bench.py --size 384 --batch 8 --dtype f16 --no-cpu > /root/repo/$O/bench_c5_384_b8_f16.json 2> /root/repo/$O/bench_c5.err
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats_c4 -- python /root/repo/bench.py --size 512 --batch 8 --no-cpu > /root/repo/$O/bench_c4_512_b8_bf16.json 2> /root/repo/$O/bench_c4.err
cd /root/repo
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
python bench.py --gpus 1 --force-split --steps 20 --warmup 5 --no-cpu --no-side > $O/bench_split.json 2> $O/bench_split.err
ls -R $O | head -40
