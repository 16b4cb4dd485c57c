#!/bin/bash
# SQ / GRBM counters + HBM traffic of the two distance-matrix kernels at config-5 size: tools/pmc_entropy_dist.sh > gpurun_out/pmc_entropy_dist.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pmced_$i
  timeout 300 rocprofv3 --pmc $ctr -d /tmp/pmced_$i -o r -- python $R/tools/entropy_dist_once.py > /tmp/pmced_$i.log 2>&1
  db=$(find /tmp/pmced_$i -name '*.db' | head -1)
  echo "== pass $i: $ctr"
  python $R/tools/pmc_summary.py $db vq_assign
done
