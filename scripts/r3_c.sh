cd /root/repo
O=gpurun_out/r3_c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "64 16 56 56 1" "64 16 28 28 3"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/t1_$tag -- python /root/repo/scripts/bench_ffn.py $cfg --reps 20 --only fused > /root/repo/$O/t1_$tag.log 2>&1
done
cd /root/repo
python bench.py --no-cpu --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/pytest.log
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
