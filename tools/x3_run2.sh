cd /root/repo
python -m pytest tests/test_gpu_train_step.py tests/test_gpu_fullsize.py "tests/test_gpu_full_configs.py::test_config1_standard_architecture_64" -q -s -k "golden or oracle or config1" 2>&1 | tail -40 > gpurun_out/x3_model_tests.log
python bench.py --dtype bf16x3 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --traffic off --sustain-s 0 --no-calibration > gpurun_out/x3_bench.json 2> gpurun_out/x3_bench.err
python bench.py --dtype f32 --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --traffic off --sustain-s 0 --no-calibration > gpurun_out/f32_bench32.json 2> gpurun_out/f32_bench32.err
tail -30 gpurun_out/x3_model_tests.log; tail -3 gpurun_out/x3_bench.err; python - <<'PY'
import json
for f in ('gpurun_out/x3_bench.json','gpurun_out/f32_bench32.json'):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, j['ms_per_step'], j['value'])
        for k,v in sorted(j['roofline']['all_kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:14]:
            print('   ',k,v)
    except Exception as e: print(f,'ERR',e)
PY
