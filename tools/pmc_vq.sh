#!/bin/bash
# SQ / GRBM counters + HBM traffic of the VQ assignment kernels at (8192, 1024, 256): tools/pmc_vq.sh > profiles/roundN_pmc_vq.txt
cd /tmp && export TMPDIR=/tmp
run="python /root/repo/tools/vq_once.py"
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_IFETCH" \
           "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVES"; do
  i=$((i+1)); rm -rf /tmp/pmcvq_$i
  rocprofv3 --pmc $ctr -d /tmp/pmcvq_$i -o r -- $run > /tmp/pmcvq_$i.log 2>&1
  db=$(find /tmp/pmcvq_$i -name '*.db' | head -1)
  echo "== pass $i: $ctr"
  python /root/repo/tools/pmc_summary.py $db vq_
done
