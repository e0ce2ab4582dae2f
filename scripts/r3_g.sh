cd /root/repo
O=gpurun_out/r3_g; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "stage_outputs or bf16_error_by_stage" -s 2>&1 | tail -15 > $O/pytest.log
timeout 900 python -m pytest tests/test_rccl_gpu.py -x -q -m gpu -s 2>&1 | tail -15 > $O/pytest_rccl.log
