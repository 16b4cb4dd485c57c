cd /root/repo
python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/full_gpu_tests.log
tail -6 gpurun_out/full_gpu_tests.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --traffic off --sustain-s 0 --no-calibration 2>/dev/null | tail -1 | cut -c1-200
python bench.py --gan --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --traffic off --sustain-s 0 --no-calibration 2>/dev/null | tail -1 | cut -c1-200
VQK_TRACE=1 python bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --traffic off --sustain-s 0 --no-calibration --no-kernel-events 2>&1 | tail -1 | cut -c1-120
