cd /root/repo
mkdir -p gpurun_out/r3_base
python bench.py --no-cpu --steps 30 --warmup 5 > gpurun_out/r3_base/bench.json 2> gpurun_out/r3_base/bench.err
for cfg in "64 16 56 56 1" "64 16 28 28 3" "128 16 14 14 3" "128 16 28 28 1" "320 16 7 7 3" "320 16 14 14 1" "512 16 7 7 1"; do
  python scripts/bench_ffn.py $cfg --reps 20 --only fused >> gpurun_out/r3_base/ffn.log 2>&1
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3_base/ffn64 -- python /root/repo/scripts/bench_ffn.py 64 16 56 56 1 --reps 20 --only fused > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3_base/ffn128 -- python /root/repo/scripts/bench_ffn.py 128 16 14 14 3 --reps 20 --only fused > /dev/null 2>&1
cd /root/repo
find gpurun_out/r3_base -name "*kernel_trace.csv" -delete
find gpurun_out/r3_base -name "*agent_info.csv" -delete
