# split-K plan of the weight-gradient GEMMs: blocks aimed for / cap (TC_SPLITK_BLOCKS / TC_SPLITK_CAP), whole-step A/B
for cfg in "512 128" "256 128" "256 64" "1024 128" "384 96" "512 128"; do set -- $cfg
  echo -n "blocks=$1 cap=$2: "; TC_SPLITK_BLOCKS=$1 TC_SPLITK_CAP=$2 python bench.py --steps 40 --warmup 5 --no-cpu --no-side 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done
