# A/B helper of round 6: a few targeted tests, two resident-batch step timings and the per-kernel averages of one traced run.
#   TESTS="tests/test_ops_gpu.py -k patchify" KERNELS="ffn_|ew_multi" bash scripts/exp/ab_step.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
if [ -n "$TESTS" ]; then timeout 1200 python -m pytest $TESTS -x -q -m gpu 2>&1 | tail -${TAILN:-6}; fi
for i in 1 2; do timeout 300 python bench.py --resident --no-cpu --no-side --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RESIDENT', round(d['value'],1), round(d['ms_per_step'],4))"; done
rm -rf gpurun_out/ab_prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/ab_prof -o t --output-format csv -- python bench.py --resident --no-cpu --no-side --steps 30 --warmup 5 > /dev/null 2>&1
python scripts/kgrep.py gpurun_out/ab_prof "${KERNELS:-ffn_|ew_multi|mhca}"
