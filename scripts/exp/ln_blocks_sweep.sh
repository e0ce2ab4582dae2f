run() { echo -n "$* : "; env "$@" python bench.py --no-side --no-cpu --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"; }
run TC_LN_BWD_BLOCKS=1024
run TC_LN_BWD_BLOCKS=2048
run TC_LN_BWD_BLOCKS=4096
run TC_LN_BWD_BLOCKS=2048 TC_LN_WG_MIN=2048
run TC_LN_BWD_BLOCKS=1024
