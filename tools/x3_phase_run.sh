#!/bin/bash
# round 6: the split-product phase forms -- parity tests, then the bf16x3 step with and without them
# (VQK_X3_PHASE: forward / data gradients, host switch; VQK_X3_WGRAD_PHASE: weight gradients, tuning slot + host switch)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv_x3.py tests/test_gpu_conv_ups_phase.py tests/test_gpu_pooled_backward.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/x3_phase_tests.log
cat gpurun_out/x3_phase_tests.log
for cfg in "1 1" "1 0" "0 0"; do
  set -- $cfg
  VQK_X3_PHASE=$1 VQK_X3_WGRAD_PHASE=$2 timeout 600 python bench.py --dtype bf16x3 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --traffic off --sustain-s 0 --no-calibration > gpurun_out/x3_phase_bench_$1$2.json 2> gpurun_out/x3_phase_bench_$1$2.err
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/x3_phase_bench_$1$2.json').read().strip().splitlines()[-1])
    print('X3_PHASE=$1 X3_WGRAD_PHASE=$2', d['ms_per_step'], d['value'], {k: v.get('ms_per_step') for k, v in d['roofline']['all_kernels'].items() if 'x3' in k})
except Exception as e:
    print('bench $1$2 failed', e)
    print(open('gpurun_out/x3_phase_bench_$1$2.err').read()[-2000:])
PY
done
