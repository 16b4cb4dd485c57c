cd /root/repo
for c in 3200 6400 12800; do echo coef $c; VQK_X3_WGRAD_COEF_E4=$c VQK_NO_FPROP=1 python tools/convbench.py x3 10 2>&1 | tail -1; done
python bench.py --dtype bf16x3 --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --traffic off --sustain-s 0 --no-calibration > gpurun_out/x3_bench.json 2> gpurun_out/x3_bench.err
python - <<'PY'
import json
for f in ('gpurun_out/x3_bench.json',):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, j['ms_per_step'], j['value'], j['step_roofline'])
        for k,v in sorted(j['roofline']['all_kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:12]:
            print('   ',k,v)
    except Exception as e: print(f,'ERR',e)
PY
python -m pytest tests/test_gpu_train_step.py tests/test_gpu_fullsize.py "tests/test_gpu_full_configs.py::test_config1_standard_architecture_64" -q -k "golden or oracle or config1" 2>&1 | tail -4
