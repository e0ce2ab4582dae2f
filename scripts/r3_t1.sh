cd /root/repo
mkdir -p gpurun_out/r3_t1
timeout 900 python -m pytest tests/test_ffn_fused_gpu.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r3_t1/pytest.log
