#!/bin/bash
# SQ / GRBM counters of one conv shape under several scratch/libvqk_<tag>.so builds: tools/pmc_mx.sh "cin cout hw k ups" tag...
cd /tmp && export TMPDIR=/tmp
shape=$1; shift
for t in "$@"; do
  for pass in 0 1; do
    if [ $pass = 0 ]; then ctr="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE";
    else ctr="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; fi
    rm -rf /tmp/pmc_$t
    VQK_LIB=/root/repo/scratch/libvqk_$t.so rocprofv3 --pmc $ctr -d /tmp/pmc_$t -o r -- python /root/repo/tools/one_conv.py $shape fprop 6 > /tmp/pmc_$t.log 2>&1
    db=$(find /tmp/pmc_$t -name '*.db' | head -1)
    echo "== $t pass $pass"
    python /root/repo/tools/pmc_summary.py $db conv3x3
  done
done
