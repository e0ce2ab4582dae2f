cd /root/repo
O=gpurun_out/r3_b; mkdir -p $O
timeout 1200 python -m pytest tests/test_ffn_fused_gpu.py -x -q -m gpu 2>&1 | tail -40 > $O/pytest.log
