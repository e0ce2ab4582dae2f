run() { echo -n "$* : "; env "$@" python bench.py --no-side --no-cpu --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))"; }
run A=1
run TC_DW_WG=512 TC_LN_WG_MIN=1024 TC_MID_WG=384
run TC_DW_WG=512 TC_LN_WG_MIN=2048 TC_MID_WG=384
run TC_DW_WG=640 TC_LN_WG_MIN=1024 TC_MID_WG=320
run TC_DW_WG=512 TC_LN_WG_MIN=1024 TC_MID_WG=256
run TC_DW_WG=512 TC_LN_WG_MIN=1024 TC_MID_WG=384 TC_LN_BWD_BLOCKS=512
run A=1
