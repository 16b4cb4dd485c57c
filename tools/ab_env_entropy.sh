#!/bin/bash
# same-box step time of config 5 (entropy K = 8192, 64 images) for several environment settings, two interleaved repetitions
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for v in "$@"; do
  ms=$(env $v python $R/bench.py --quantizer entropy --codebook 8192 --batch 64 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --no-other-configs --traffic off --sustain-s 0 --no-calibration 2>/dev/null < /dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')
  echo "$v  $ms"
done; done
