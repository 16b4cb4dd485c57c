#!/bin/bash
# Round 6, after the split-product phase forms: the bf16x3 part of tools/collect_round6.sh again (one box) + the whole GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/full_gpu_tests.log
python bench.py --dtype bf16x3 --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --traffic off --sustain-s 0 --no-calibration 2>/dev/null | tail -1 > gpurun_out/round6_bench_bf16x3.json
tools/cfg_kt.sh round6_bf16x3 --dtype bf16x3 --batch 32 --no-calibration
python bench.py --steps 20 --warmup 5 --quick 2>/dev/null | tail -1 > gpurun_out/round6_bench_after_phase.json
cat gpurun_out/full_gpu_tests.log
python - <<'PY'
import json
for f in ('round6_bench_bf16x3', 'round6_bench_after_phase'):
    j = json.loads(open(f'gpurun_out/{f}.json').read().strip().splitlines()[-1])
    print(f, j['ms_per_step'], j['value'], j['roofline'].get('frac'))
    for o in j.get('other_configs', []) if isinstance(j.get('other_configs'), list) else []:
        print('   ', o.get('name') or o.get('config'), o.get('ms_per_step'), o.get('img_s') or o.get('value'))
PY
