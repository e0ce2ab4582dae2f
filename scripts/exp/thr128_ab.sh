# 128x128 GEMM tiles from `thr` tiles on (TC_GEMM_THR128; default 100000 = never) -- whole step
for t in 100000 4000 1000 300 100000; do
  TC_GEMM_THR128=$t python bench.py --no-side --no-cpu --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('thr128=$t', round(d['value'],1), round(d['ms_per_step'],3))"
done
