# one-at-a-time re-sweep of the launch-geometry switches at HEAD (whole step, ms)
run() { echo -n "$* : "; env "$@" python bench.py --no-side --no-cpu --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"; }
run A=1
run TC_DKV_SLOTS=384
run TC_DKV_SLOTS=768
run TC_BN_ROWS_PER_WG=8
run TC_BN_ROWS_PER_WG=32
run TC_BN_BWD_ROWS_PER_WG=8
run TC_BN_BWD_ROWS_PER_WG=32
run TC_BN_CHUNKS=64
run TC_DW_WGRAD3_WG=512
run TC_FFN_MID_THREADS=256
run TC_SPLITK_CAP=64
run TC_SPLITK_CAP=256
run TC_FFN_LN_GEMM_MAXC=0
run TC_FFN_STORE_ACT=0
run A=1
