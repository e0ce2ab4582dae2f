# round-3 baseline: GPU tests, default bench, per-kernel trace of single MixFFN sites (tiled forward on / off)
cd /root/repo
O=gpurun_out/r3_a; mkdir -p $O
python bench.py --no-cpu --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for cfg in "64 16 56 56 1" "64 16 28 28 3" "128 16 28 28 1" "128 16 14 14 3"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/t1_$tag -- python /root/repo/scripts/bench_ffn.py $cfg --reps 20 --only fused > /root/repo/$O/t1_$tag.log 2>&1
  TC_FFN_TILED=0 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/t0_$tag -- python /root/repo/scripts/bench_ffn.py $cfg --reps 20 --only fused > /root/repo/$O/t0_$tag.log 2>&1
done
cd /root/repo
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
