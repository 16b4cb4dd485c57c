cd /root/repo
python -m pytest tests/test_gpu_conv_x3.py -q 2>&1 | tail -25 > gpurun_out/x3_test.log
tail -12 gpurun_out/x3_test.log
