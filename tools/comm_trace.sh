#!/bin/bash
# kernel trace of the stand-in step (tools/comm_probe.py) per TILE_QUEUE mode -> gpurun_out/<tag>_m<mode>_{timeline.txt,csv}
# usage: tools/comm_trace.sh tag [blocks=16] [ms=1.5]
tag=${1:-ct}; blocks=${2:-16}; ms=${3:-1.5}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for mode in 0 2; do
  rm -rf /tmp/prof_${tag}_$mode
  rocprofv3 --kernel-trace -d /tmp/prof_${tag}_$mode -o p -- python $R/tools/comm_probe.py --steps 6 --modes $mode --blocks $blocks --ms $ms > $R/gpurun_out/${tag}_m$mode.log 2>&1
  db=$(find /tmp/prof_${tag}_$mode -name '*.db' | head -1)
  TL_MARK=probe python $R/tools/timeline.py $db > $R/gpurun_out/${tag}_m${mode}_timeline.txt 2>&1
  python $R/tools/rocpd_stats.py $db $R/gpurun_out/${tag}_m$mode.csv > /dev/null 2>&1
done
