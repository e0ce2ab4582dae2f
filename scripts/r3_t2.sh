cd /root/repo
mkdir -p gpurun_out/r3_t2
python bench.py --no-cpu --steps 30 --warmup 5 > gpurun_out/r3_t2/bench.json 2> gpurun_out/r3_t2/bench.err
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3_t2/pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3_t2/ffn64 -- python /root/repo/scripts/bench_ffn.py 64 16 56 56 1 --reps 20 --only fused > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3_t2/ffn64g -- python /root/repo/scripts/bench_ffn.py 64 16 28 28 3 --reps 20 --only fused > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3_t2/ffn128 -- python /root/repo/scripts/bench_ffn.py 128 16 14 14 3 --reps 20 --only fused > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3_t2/ffn128b -- python /root/repo/scripts/bench_ffn.py 128 16 28 28 1 --reps 20 --only fused > /dev/null 2>&1
cd /root/repo
find gpurun_out/r3_t2 -name "*kernel_trace.csv" -delete; find gpurun_out/r3_t2 -name "*agent_info.csv" -delete
