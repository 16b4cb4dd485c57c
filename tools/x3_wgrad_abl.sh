#!/bin/bash
# timing-only ablations of conv3x3_wgrad_x3_kernel (csrc/conv_x3.hip, VQK_X3ABL): which resource bounds it?
# build the variants first (here, before gpurun):  for v in 1 2 4 8 3; do AB_SRC=conv_x3.hip tools/ab_build.sh x3abl$v -DVQK_X3ABL=$v; done
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/x3_wgrad_abl.txt; : > $out
for v in base x3abl1 x3abl2 x3abl4 x3abl8 x3abl3; do
  echo "== $v" >> $out
  if [ $v = base ]; then lib=""; else lib="$PWD/ab_libs/libvqk_$v.so"; fi
  VQK_LIB=$lib VQK_NO_FPROP=1 timeout 300 python tools/convbench.py x3 20 2>&1 | grep -v "^variant" >> $out
done
cat $out
