cd /root/repo
python -m pytest tests/test_gpu_two_models.py tests/test_gpu_gan.py tests/test_gpu_full_configs.py -q -x 2>&1 | tail -8
