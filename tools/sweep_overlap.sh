#!/bin/bash
# same-box sweep of the wgrad side-stream cap / overlap mode: tools/sweep_overlap.sh
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events --no-other-configs --traffic off --sustain-s 0"
ms() { "$@" 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for rep in 1 2; do
for cap in 192 256 320 384 448 512; do echo -n "cap $cap: "; VQK_OVERLAP_WGRAD_BLOCKS=$cap ms $B; done
for mode in 1 2 3; do echo -n "mode $mode: "; VQK_OVERLAP_MODE=$mode ms $B; done
done
