#!/bin/bash
# rocprofv3 kernel-trace summary of the VQ-GAN step (config 4): per-kernel totals -> gpurun_out/gan_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_gan
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_gan -o p -- python $R/bench.py --gan --batch 16 --steps 6 --warmup 3 --no-graph \
  --no-cpu-baseline --no-kernel-events --no-other-configs --traffic off --sustain-s 0 --no-calibration > $R/gpurun_out/gan_kt.log 2>&1 < /dev/null
python $R/tools/rocpd_stats.py /tmp/kt_gan/p_results.db $R/gpurun_out/gan_kernel_stats.csv < /dev/null
python $R/tools/rocpd_seq.py /tmp/kt_gan/p_results.db "thin_in_kernel<4>" 30 < /dev/null
