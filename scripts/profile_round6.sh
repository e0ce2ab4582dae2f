# round-6 profile set (ONE gpurun call at the end of the round: VERDICT r5 item 9).  Every summary is stamped with the sha256 of the sources it was collected from
# (scripts/provenance.py stamp); bench.py reports a profile's traffic only while that stamp matches the sources it runs.
#   attention counters (4 passes) -> r6_attn_pmc.json, r6_attn_kernel_stats.csv
#   per-kernel HBM traffic + timeline of one replayed (resident-batch) step (3 passes) -> r6_hbm_by_kernel.json, r6_step_timeline.json
#   section timeline of the same step -> r6_seg_timeline.json / .txt
#   RIPM / IFF stages alone (3 passes each) -> r6_ripm_iff_hbm.json
#   kernel stats + bench line of the default command, of BASELINE configs 4 / 5, the split-step bench
cd /root/repo
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6_final; rm -rf $O; mkdir -p $O
bash scripts/pmc_attn.sh pmc_attn_r6 > gpurun_out/pmc_attn_r6.log 2>&1
python scripts/pmc_summary.py gpurun_out/pmc_attn_r6/attn_pmc.json gpurun_out/pmc_attn_r6/fetch gpurun_out/pmc_attn_r6/write gpurun_out/pmc_attn_r6/sq > gpurun_out/pmc_attn_r6/summary.txt 2>&1
cp $(find gpurun_out/pmc_attn_r6/trace -name "*kernel_stats.csv" | head -1) $O/r6_attn_kernel_stats.csv
cp gpurun_out/pmc_attn_r6/attn_pmc.json $O/r6_attn_pmc.json
find gpurun_out/pmc_attn_r6 -name "*kernel_trace.csv" -delete; find gpurun_out/pmc_attn_r6 -name "*counter_collection.csv" -delete; find gpurun_out/pmc_attn_r6 -name "*agent_info.csv" -delete
bash scripts/pmc_step.sh pmc_step_r6 > gpurun_out/pmc_step_r6.log 2>&1
python scripts/hbm_by_kernel.py gpurun_out/pmc_step_r6 $O/r6_hbm_by_kernel.json > gpurun_out/pmc_step_r6/hbm.txt 2>&1
python scripts/step_profile.py $(find gpurun_out/pmc_step_r6/trace -name "*kernel_trace.csv" | head -1) --json $O/r6_step_timeline.json > gpurun_out/pmc_step_r6/step_profile.txt 2>&1
find gpurun_out/pmc_step_r6 -name "*kernel_trace.csv" -delete; find gpurun_out/pmc_step_r6 -name "*counter_collection.csv" -delete; find gpurun_out/pmc_step_r6 -name "*agent_info.csv" -delete
bash scripts/seg_profile.sh seg_r6 > /dev/null 2>&1
cp gpurun_out/seg_r6/seg_timeline.json $O/r6_seg_timeline.json; cp gpurun_out/seg_r6/seg_timeline.txt $O/r6_seg_timeline.txt
bash scripts/pmc_stage.sh pmc_stage_r6 > gpurun_out/pmc_stage_r6.log 2>&1
cp gpurun_out/pmc_stage_r6/ripm_iff_hbm.json $O/r6_ripm_iff_hbm.json
python scripts/provenance.py stamp $O/r6_attn_pmc.json $O/r6_hbm_by_kernel.json $O/r6_step_timeline.json $O/r6_seg_timeline.json $O/r6_ripm_iff_hbm.json > $O/provenance.log 2>&1
cp $O/r6_*.json profiles/ 2>/dev/null     # (on the box: so that the bench lines below see the fresh, stamped profiles)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats -- python /root/repo/bench.py > /root/repo/$O/bench.json 2> /root/repo/$O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats_c5 -- python /root/repo/bench.py --size 384 --batch 8 --dtype f16 --no-cpu > /root/repo/$O/bench_c5_384_b8_f16.json 2> /root/repo/$O/bench_c5.err
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats_c4 -- python /root/repo/bench.py --size 512 --batch 8 --no-cpu > /root/repo/$O/bench_c4_512_b8_bf16.json 2> /root/repo/$O/bench_c4.err
cd /root/repo
for d in stats stats_c5 stats_c4; do cp $(find $O/$d -name "*kernel_stats.csv" | head -1) $O/${d}_kernel_stats.csv 2>/dev/null; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
python bench.py --gpus 1 --force-split --steps 20 --warmup 5 --no-cpu --no-side > $O/bench_split.json 2> $O/bench_split.err
ls -R $O | head -40
# round-6 evidence that is not rocprofv3 output: the bridge attention under sustained load (clock / power from rocm-smi beside the launch
# time; random and all-zero operands, forward and backward call) -- why the streams sit where they do against the 2.4 GHz MFMA peak
for a in "" "--zeros" "--bwd" "--bwd --zeros"; do python scripts/exp/attn_power.py $a 2>&1 | grep sustained; done > $O/r6_attn_sustained_power.txt
ls $O
