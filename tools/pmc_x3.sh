#!/bin/bash
# SQ / GRBM / TCC counters of the split-product kernels (csrc/conv_x3.hip) on ONE shape, separate --pmc passes:
#   tools/pmc_x3.sh "cin cout hw k ups" > gpurun_out/round6_pmc_x3.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
shape=${1:-"128 128 256 3 0"}
export VQK_ONE_CONV_MODE=x3
for which in fprop wgrad; do
  for pass in 0 1 2 3; do
    case $pass in
      0) ctr="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE";;
      1) ctr="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE";;
      2) ctr="FETCH_SIZE";;
      3) ctr="WRITE_SIZE";;
    esac
    rm -rf /tmp/pmc_x3
    rocprofv3 --pmc $ctr -d /tmp/pmc_x3 -o r -- python $R/tools/one_conv.py $shape $which 6 > /tmp/pmc_x3.log 2>&1
    db=$(find /tmp/pmc_x3 -name '*.db' | head -1)
    echo "== $which ($shape, 32 images) pass $pass: $ctr"
    python $R/tools/pmc_summary.py $db x3_kernel
  done
done
