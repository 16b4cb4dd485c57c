#!/bin/bash
# named ranges of one EAGER train step (VQK_TRACE=1: roctx ranges around every macro-op of ops.py) next to its kernels:
# rocprofv3 --marker-trace --kernel-trace --stats (no counters) -> gpurun_out/<tag>_marker_stats.csv, <tag>_marker_kernel_stats.csv
# usage: tools/trace_step.sh tag [bench args]
tag=${1:-trace}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mt_$tag
VQK_TRACE=1 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d /tmp/mt_$tag -o p -- python $R/bench.py --no-graph --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events --no-other-configs --traffic off --sustain-s 0 --no-calibration "$@" > $R/gpurun_out/${tag}_marker.log 2>&1
ls -R /tmp/mt_$tag | head -30 >> $R/gpurun_out/${tag}_marker.log
f=$(find /tmp/mt_$tag -name '*marker_api_stats.csv' | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/${tag}_marker_stats.csv
f=$(find /tmp/mt_$tag -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/${tag}_marker_kernel_stats.csv
f=$(find /tmp/mt_$tag -name '*marker_api_trace.csv' | head -1); [ -n "$f" ] && head -400 $f > $R/gpurun_out/${tag}_marker_trace_head.csv
