# A/B of two libraries on the bridge attention micro-benchmark, interleaved rounds: bash scripts/attn_ab.sh libA.so libB.so (names under transception_amd/)
cd /root/repo
O=gpurun_out/attn_ab; rm -rf $O; mkdir -p $O
for r in 1 2 3; do
  for lib in "$@"; do
    echo "== $lib" >> $O/ab.log
    TC_LIB_PATH=transception_amd/$lib python scripts/bench_attn.py 2>/dev/null | grep attn >> $O/ab.log
  done
done
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attn" 2>&1 | tail -3 >> $O/ab.log
