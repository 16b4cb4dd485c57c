#!/bin/bash
# rocprofv3 kernel-trace summary of any bench configuration: tools/cfg_kt.sh <tag> <bench args...> -> gpurun_out/<tag>_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$tag
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o p -- python $R/bench.py "$@" --steps 6 --warmup 3 --no-graph \
  --no-cpu-baseline --no-kernel-events --no-other-configs --traffic off --sustain-s 0 > $R/gpurun_out/${tag}_kt.log 2>&1 < /dev/null
python $R/tools/rocpd_stats.py /tmp/kt_$tag/p_results.db $R/gpurun_out/${tag}_kernel_stats.csv < /dev/null
