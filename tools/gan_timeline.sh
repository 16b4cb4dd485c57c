#!/bin/bash
# kernel-trace timeline of the REPLAYED VQ-GAN step (config 4): the AdamW-to-AdamW intervals = the AE half and the D half
#   -> gpurun_out/gan_timeline.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_gant
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_gant -o p -- python $R/bench.py --gan --batch 16 --steps 10 --warmup 4 \
  --no-cpu-baseline --no-kernel-events --no-other-configs --traffic off --sustain-s 0 --no-calibration > $R/gpurun_out/gan_timeline.log 2>&1 < /dev/null
for back in 2 3 4 5; do echo "=== interval ending at AdamW launch -$back"; python $R/tools/timeline.py /tmp/kt_gant/p_results.db $back 3 < /dev/null | head -${TL_HEAD:-34}; done > $R/gpurun_out/gan_timeline.txt 2>&1
