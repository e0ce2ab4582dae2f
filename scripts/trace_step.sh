set -u
OUT=gpurun_out/${1:-trace_step}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python bench.py --resident --no-cpu --no-side --steps 3 --warmup 1 > $OUT/trace.log 2>&1
