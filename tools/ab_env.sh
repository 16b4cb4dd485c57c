#!/bin/bash
# same-box A/B of the train step with / without an environment switch: tools/ab_env.sh VAR=VALUE [bench args]
kv=$1; shift
for rep in 1 2; do for t in off on; do
  if [ $t = on ]; then export "$kv"; else unset "${kv%%=*}"; fi
  echo -n "$kv $t: "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['final_loss'])"
done; done
