#!/bin/bash
# HBM traffic per kernel of the benchmarked step (VERDICT r1 item 5 / SURVEY 8(d)): three passes of the same short bench command --
# a clean kernel trace for durations, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes (they do not fit one pass, and gpurun
# refuses --pmc together with the API trace domains).  Output under gpurun_out/$1; fold with scripts/hbm_by_kernel.py.
set -u
OUT=gpurun_out/${1:-pmc_step}
CMD="python bench.py --resident --no-cpu --no-side --steps 3 --warmup 1 ${BENCH_ARGS:-}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- $CMD > $OUT/write.log 2>&1
ls -R $OUT | head -20
