#!/bin/bash
# Counters of the bridge attention kernels at the bench shape (scripts/bench_attn.py): a clean trace, FETCH_SIZE, WRITE_SIZE and the SQ
# group in separate --pmc passes.  Output under gpurun_out/$1; fold with scripts/pmc_summary.py.
set -u
OUT=gpurun_out/${1:-pmc_attn}
CMD="python scripts/bench_attn.py"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- $CMD > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq -o s -- $CMD > $OUT/sq.log 2>&1
cat $OUT/trace.log | tail -3
ls -R $OUT | head -30
