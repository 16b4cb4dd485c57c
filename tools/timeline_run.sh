#!/bin/bash
# kernel-trace of a short bench run + the queue-level timeline of one replayed step -> gpurun_out/<name>_timeline.txt
# usage: [ENV=...] tools/timeline_run.sh name [bench args]
name=${1:-tl}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$name
rocprofv3 --kernel-trace -d /tmp/prof_$name -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-other-configs --traffic off --sustain-s 0 "$@" > $R/gpurun_out/$name.log 2>&1
db=$(find /tmp/prof_$name -name '*.db' | head -1)
python $R/tools/timeline.py $db > $R/gpurun_out/${name}_timeline.txt 2>&1
python $R/tools/rocpd_stats.py $db $R/gpurun_out/$name.csv > /dev/null 2>&1
