#!/bin/bash
# step time for several values of one environment variable: tools/sweep_env.sh VAR v1 v2 ... (two interleaved repetitions)
var=$1; shift
for rep in 1 2; do for v in "$@"; do
  export $var=$v
  echo -n "$var=$v: "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done; done
