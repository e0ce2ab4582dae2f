cd /root/repo
O=gpurun_out/r3_g; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_data_gpu.py tests/test_trainer_gpu.py -x -q -m gpu 2>&1 | tail -15 > $O/pytest.log
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "overflow" 2>&1 | tail -8 >> $O/pytest.log
python __graft_entry__.py --smoke 2>&1 | tail -4 >> $O/pytest.log
