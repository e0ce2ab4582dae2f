cd /root/repo
O=gpurun_out/step_ab; rm -rf $O; mkdir -p $O
python bench.py --no-cpu --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err
TC_FFN_TILED_BWD=0 python bench.py --no-cpu --no-side --steps 30 --warmup 5 > $O/bench_nobwd.json 2> $O/bench_nobwd.err
timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/pytest.log
