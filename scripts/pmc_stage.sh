#!/bin/bash
# Kernel time and memory-side bytes of the RIPM and IFF stages alone (scripts/bench_stage.py), three rocprofv3 passes each (trace,
# FETCH_SIZE, WRITE_SIZE); scripts/stage_summary.py folds them into profiles/r4_ripm_iff_hbm.json.
set -u
OUT=gpurun_out/${1:-pmc_stage}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
for st in ripm iff; do
  CMD="python scripts/bench_stage.py $st --iters 10"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$st/trace -o t -- $CMD > $OUT/$st.json 2> $OUT/$st.trace.log
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$st/fetch -o f -- $CMD > /dev/null 2> $OUT/$st.fetch.log
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/$st/write -o w -- $CMD > /dev/null 2> $OUT/$st.write.log
done
python scripts/stage_summary.py $OUT $OUT/ripm_iff_hbm.json
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
cat $OUT/ripm_iff_hbm.json
