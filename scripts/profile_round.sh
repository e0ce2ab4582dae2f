# round-3 profile set: per-kernel HBM traffic of one step (3 passes), kernel stats of the default bench command, the bench line itself
cd /root/repo
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash scripts/pmc_step.sh pmc_step_round > gpurun_out/pmc_step_round.log 2>&1
python scripts/hbm_by_kernel.py gpurun_out/pmc_step_round gpurun_out/pmc_step_round/hbm_by_kernel.json > gpurun_out/pmc_step_round/hbm.txt 2>&1
python scripts/step_profile.py $(find gpurun_out/pmc_step_round/trace -name "*kernel_trace.csv" | head -1) --json gpurun_out/pmc_step_round/step_timeline.json > gpurun_out/pmc_step_round/step_profile.txt 2>&1
find gpurun_out/pmc_step_round -name "*kernel_trace.csv" -delete; find gpurun_out/pmc_step_round -name "*counter_collection.csv" -delete; find gpurun_out/pmc_step_round -name "*agent_info.csv" -delete
O=gpurun_out/round_final; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats -- python /root/repo/bench.py > /root/repo/$O/bench.json 2> /root/repo/$O/bench.err
cd /root/repo
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
python bench.py --gpus 1 --force-split --steps 20 --warmup 5 --no-cpu --no-side > $O/bench_split.json 2> $O/bench_split.err
