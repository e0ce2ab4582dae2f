# rows per workgroup of the BatchNorm apply passes (forward / backward): RIPM stages alone and the whole step
for cfg in "16 16" "16 64" "64 64" "16 32" "32 32" "16 16"; do set -- $cfg
  echo -n "fwd=$1 bwd=$2: ripm "; TC_BN_ROWS_PER_WG=$1 TC_BN_BWD_ROWS_PER_WG=$2 python scripts/bench_stage.py ripm 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['us_per_fwd_bwd'],1), end=' ')"
  echo -n " step "; TC_BN_ROWS_PER_WG=$1 TC_BN_BWD_ROWS_PER_WG=$2 python bench.py --steps 40 --warmup 5 --no-cpu --no-side 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"
done
