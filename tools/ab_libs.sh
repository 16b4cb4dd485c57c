#!/bin/bash
# same-box step time for several scratch/libvqk_<tag>.so builds, two interleaved repetitions: tools/ab_libs.sh tagA tagB ...
for rep in 1 2; do for t in "$@"; do
  echo -n "$t: "; VQK_LIB=/root/repo/scratch/libvqk_$t.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done; done
