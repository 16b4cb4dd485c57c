#!/bin/bash
# same-box A/B of one environment variable over the headline step: tools/ab_env_step.sh VAR v1 v2 ...  (each value twice, interleaved)
R=${GRAFT_REPO_ROOT:-/root/repo}
var=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    ms=$(env $var=$v python $R/bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-events --no-other-configs --traffic off --sustain-s 0 --no-calibration 2>/dev/null < /dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')
    echo "$var=$v  $ms"
  done
done
