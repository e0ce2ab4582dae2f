# walker / block counts again now that the fold tails are gone (deferred folds): TC_DW_WG (depthwise weight-gradient walkers per launch), TC_LN_WG_MIN
run() { echo -n "$* : "; env "$@" python bench.py --no-side --no-cpu --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))"; }
run A=1
run TC_DW_WG=384
run TC_DW_WG=512
run TC_DW_WG=768
run TC_DW_WG=192
run TC_LN_WG_MIN=1024
run TC_LN_WG_MIN=256
run TC_MID_WG=384
run TC_MID_WG=512
run A=1
