#!/bin/bash
# kernel-trace of a short bench run -> gpurun_out/<name>.csv (per-kernel stats); usage: tools/prof_step.sh name [bench args]
name=${1:-prof}; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof_$name -o p -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events "$@" > /root/repo/gpurun_out/$name.log 2>&1
python /root/repo/tools/rocpd_stats.py /tmp/prof_$name/p_results.db /root/repo/gpurun_out/$name.csv
