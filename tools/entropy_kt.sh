#!/bin/bash
# rocprofv3 kernel-trace summary of config 5 (entropy quantizer, K = 8192, 64 images): the quantizer's kernels -> gpurun_out/entropy_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_ent
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_ent -o p -- python $R/bench.py --quantizer entropy --codebook 8192 --batch 64 --steps 6 --warmup 3 \
  --no-cpu-baseline --no-kernel-events --no-other-configs --traffic off --sustain-s 0 --no-calibration > $R/gpurun_out/entropy_kt.log 2>&1 < /dev/null
python $R/tools/rocpd_stats.py /tmp/kt_ent/p_results.db $R/gpurun_out/entropy_kernel_stats.csv < /dev/null
