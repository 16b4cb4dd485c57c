#!/bin/bash
# rocprofv3 kernel-trace summary of the fp32 parity mode (bs 8) -> gpurun_out/fp32_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_f32
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_f32 -o p -- python $R/bench.py --dtype f32 --batch 8 --steps 6 --warmup 3 \
  --no-cpu-baseline --no-kernel-events --no-other-configs --traffic off --sustain-s 0 --no-calibration > $R/gpurun_out/fp32_kt.log 2>&1 < /dev/null
python $R/tools/rocpd_stats.py /tmp/kt_f32/p_results.db $R/gpurun_out/fp32_kernel_stats.csv < /dev/null
