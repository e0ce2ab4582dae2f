cd /root/repo
O=gpurun_out/r3_f; rm -rf $O; mkdir -p $O
CMD="python bench.py --no-cpu --no-side --steps 3 --warmup 1"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- $CMD > $O/trace.log 2>&1
python scripts/step_profile.py $(find $O/trace -name "*kernel_trace.csv" | head -1) --json $O/step_timeline.json > $O/step_profile.txt 2>&1
TC_FFN_TILED_BWD=0 rocprofv3 --kernel-trace --output-format csv -d $O/trace0 -o t -- $CMD > $O/trace0.log 2>&1
python scripts/step_profile.py $(find $O/trace0 -name "*kernel_trace.csv" | head -1) --json $O/step_timeline0.json > $O/step_profile0.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
