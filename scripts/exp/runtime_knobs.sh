run() { echo "== $*"; env "$@" python bench.py --no-side --no-cpu --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
run A=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_HIP_GRAPH_DOT_PRINT=0 HIP_FORCE_DEV_KERNARG=1 GPU_MAX_HW_QUEUES=1
run AMD_DIRECT_DISPATCH=1
run A=1
rocm-smi --showperflevel --showclocks --showpower 2>/dev/null | grep -v "^$" | head -30
