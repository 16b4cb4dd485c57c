#!/bin/bash
# same-box VQ-GAN step time (config 4, bs 16) for several environment settings, two interleaved repetitions:
#   tools/ab_env_gan.sh "A=1 B=2" "A=3" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for v in "$@"; do
  ms=$(env $v python $R/bench.py --gan --batch 16 --steps 32 --warmup 8 --no-cpu-baseline --no-kernel-events --no-other-configs --traffic off --sustain-s 0 --no-calibration 2>/dev/null < /dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')
  echo "$v  $ms"
done; done
