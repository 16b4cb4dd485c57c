cd /root/repo
python -m pytest tests/test_gpu_thin_f32.py -q 2>&1 | tail -5
python -m pytest "tests/test_gpu_train_step.py::test_train_step_fp32_golden" -q -k "ema" 2>&1 | grep -E "Mismatch|Max|assert|Error|passed|failed" | head -20
python bench.py --dtype bf16x3 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --traffic off --sustain-s 0 --no-calibration > gpurun_out/x3_bench.json 2> gpurun_out/x3_bench.err
python - <<'PY'
import json
for f in ('gpurun_out/x3_bench.json',):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, j['ms_per_step'], j['value'])
        for k,v in sorted(j['roofline']['all_kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:16]:
            print('   ',k,v)
    except Exception as e: print(f,'ERR',e)
PY
