cd /root/repo
O=gpurun_out/ffn_sites; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_ffn_fused_gpu.py -x -q -m gpu 2>&1 | tail -5 > $O/pytest.log
for cfg in "64 16 56 56 1" "128 16 28 28 1" "128 16 14 14 3"; do
  TC_LIB_PATH=transception_amd/libtc_ffnb.so python scripts/exp/ffnb_timing.py $cfg >> $O/timing.log 2>&1
done
cd /tmp && export TMPDIR=/tmp
for cfg in "64 16 56 56 1" "64 16 28 28 3" "128 16 28 28 1" "128 16 14 14 3"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/t1_$tag -- python /root/repo/scripts/bench_ffn.py $cfg --reps 20 --only fused > /root/repo/$O/t1_$tag.log 2>&1
done
cd /root/repo
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
