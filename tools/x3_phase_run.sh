#!/bin/bash
# round 6: the split-product phase forms -- parity tests, then the bf16x3 step with and without them
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv_x3.py tests/test_gpu_conv_ups_phase.py tests/test_gpu_pooled_backward.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/x3_phase_tests.log
cat gpurun_out/x3_phase_tests.log
for ph in 1 0; do
  VQK_X3_PHASE=$ph timeout 600 python bench.py --dtype bf16x3 --batch 32 --steps 10 --warmup 3 --quick > gpurun_out/x3_phase_bench_$ph.json 2> gpurun_out/x3_phase_bench_$ph.err
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/x3_phase_bench_$ph.json').read().strip().splitlines()[-1])
    print('X3_PHASE=$ph', d['ms_per_step'], d['value'])
except Exception as e:
    print('bench $ph failed', e)
    print(open('gpurun_out/x3_phase_bench_$ph.err').read()[-2000:])
PY
done
