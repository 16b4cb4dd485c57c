#!/bin/bash
# per-shape fprop time of several scratch/libvqk_<tag>.so builds on the large-map shapes: tools/ab_mx.sh tag...
# (tag "stream" = the current build with the matrix/auxiliary-wave kernel switched off)
for t in "$@"; do
  echo "== $t"
  if [ "$t" = stream ]; then lib=/root/repo/scratch/libvqk_cur.so; mx=0; else lib=/root/repo/scratch/libvqk_$t.so; mx=1; fi
  VQK_MX=$mx VQK_LIB=$lib VQK_NO_WGRAD=1 timeout 200 python /root/repo/tools/convbench.py bf16 10 2>&1 | head -11 | tail -8 | cut -c1-66
done
