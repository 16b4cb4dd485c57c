cd /root/repo
python bench.py --dtype bf16x3 --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --traffic off --sustain-s 0 --no-calibration > gpurun_out/x3_bench.json 2> gpurun_out/x3_bench.err
python - <<'PY'
import json
for f in ('gpurun_out/x3_bench.json',):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, j['ms_per_step'], j['value'])
        for k,v in sorted(j['roofline']['all_kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:12]:
            print('   ',k,v)
    except Exception as e: print(f,'ERR',e)
PY
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/full_gpu_tests.log
tail -5 gpurun_out/full_gpu_tests.log
