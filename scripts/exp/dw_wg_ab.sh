# tile walkers of the depthwise weight-gradient / MixFFN middle kernels (TC_DW_WG / TC_MID_WG): whole-step A/B
for cfg in "256 256" "512 256" "1024 256" "256 512" "512 512" "256 256"; do set -- $cfg
  echo -n "dw=$1 mid=$2: "; TC_DW_WG=$1 TC_MID_WG=$2 python bench.py --steps 40 --warmup 5 --no-cpu --no-side 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"
done
