#!/bin/bash
# per-shape wgrad time of scratch/libvqk_<tag>.so builds: tools/ab_wg.sh tag...   ("old" = current build, single-role kernel)
for t in "$@"; do
  echo "== $t"
  if [ "$t" = old ]; then lib=/root/repo/scratch/libvqk_cur.so; mx=0; else lib=/root/repo/scratch/libvqk_$t.so; mx=1; fi
  VQK_WGMX=$mx VQK_LIB=$lib VQK_NO_FPROP=1 timeout 300 python /root/repo/tools/convbench.py bf16 10 2>&1 | head -20 | tail -17 | cut -c1-38,66-90
done
