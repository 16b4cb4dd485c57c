cd /root/repo
python -m pytest tests/test_gpu_status.py tests/test_gpu_deterministic.py tests/test_gpu_two_models.py tests/test_gpu_tile_queue.py tests/test_gpu_dist.py tests/test_gpu_train_step.py -q -x 2>&1 | tail -6
