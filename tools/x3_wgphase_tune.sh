#!/bin/bash
# tuning of the phase-form x3 weight gradient: block cap (a third block per CU fits) and split coefficient, in the bf16x3 step
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/x3_wgphase_tune.txt; : > $out
for cfg in "100 3200" "150 3200" "100 1600" "100 6400" "150 6400" "125 3200"; do
  set -- $cfg
  VQK_X3_WGRAD_PHASE_CAP_PCT=$1 VQK_X3_WGRAD_PHASE_COEF_E4=$2 timeout 600 python bench.py --dtype bf16x3 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-kernel-events --traffic off --sustain-s 0 --no-calibration 2>/dev/null | tail -1 > gpurun_out/tmp_tune.json
  python - <<PY >> $out
import json
d = json.loads(open('gpurun_out/tmp_tune.json').read().strip().splitlines()[-1])
print('cap_pct $1 coef_e4 $2', d['ms_per_step'], d['value'])
PY
done
cat $out
