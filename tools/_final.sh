cd /root/repo
python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/final_gpu_tests.log
tail -2 gpurun_out/final_gpu_tests.log
bash tools/collect_round6.sh
