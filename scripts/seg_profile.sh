#!/bin/bash
# section timeline of one replayed step (scripts/seg_timeline.py): kernel trace of a short bench run with TC_SEG_MARKS on
set -u
OUT=gpurun_out/${1:-seg}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
TC_SEG_MARKS=$GRAFT_REPO_ROOT/$OUT/labels.json rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python bench.py --resident --no-cpu --no-side --steps 3 --warmup 1 ${BENCH_ARGS:-} > $OUT/trace.log 2>&1
python scripts/seg_timeline.py $OUT/trace $OUT/labels.json --json $OUT/seg_timeline.json --top 14 > $OUT/seg_timeline.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
tail -5 $OUT/trace.log
head -40 $OUT/seg_timeline.txt
