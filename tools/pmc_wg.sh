#!/bin/bash
# SQ / GRBM counters of one wgrad shape: tools/pmc_wg.sh "cin cout hw k ups" tag...  ("old": single-role kernel of the current build)
cd /tmp && export TMPDIR=/tmp
shape=$1; shift
for t in "$@"; do
  if [ "$t" = old ]; then lib=/root/repo/scratch/libvqk_cur.so; mx=0; else lib=/root/repo/scratch/libvqk_$t.so; mx=1; fi
  rm -rf /tmp/pmc_$t
  VQK_WGMX=$mx VQK_LIB=$lib rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE -d /tmp/pmc_$t -o r -- python /root/repo/tools/one_conv.py $shape wgrad 6 > /tmp/pmc_$t.log 2>&1
  db=$(find /tmp/pmc_$t -name '*.db' | head -1)
  echo "== $t"
  python /root/repo/tools/pmc_summary.py $db wgrad
  python /root/repo/tools/rocpd_stats.py $db 2>/dev/null | grep -i wgrad | head -3
done
