#!/bin/bash
# same-box A/B of the number of HSA hardware queues the HIP runtime may create (GPU_MAX_HW_QUEUES, default 4): a replayed hipGraph
# spreads its branches over all of them.  usage: tools/ab_hwq.sh "1 2 3 4 8" [bench args]
qs=${1:-"2 3 4"}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for q in $qs; do
    out=$(GPU_MAX_HW_QUEUES=$q python $R/bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events --no-other-configs --traffic off --sustain-s 0 --no-calibration "$@" 2>/dev/null | tail -1)
    echo "GPU_MAX_HW_QUEUES=$q rep $rep: $(echo "$out" | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print(j["ms_per_step"], "ms/step")')"
  done
done
