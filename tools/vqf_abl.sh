#!/bin/bash
# timing-only ablations of the VQ filter kernel (scratch/libvqk_vqf<bits>.so): bits 1 no re-rank, 2 no pass 2, 4 no pass 1, 8 no z staging
for t in "" vqf1 vqf3 vqf7 vqf15; do
  if [ -z "$t" ]; then lib=""; else lib=/root/repo/scratch/libvqk_$t.so; fi
  echo -n "${t:-full}: "; VQK_LIB=$lib python -c "
import sys; sys.path.insert(0,'/root/repo/tools'); import vqbench
r=vqbench.bench(8192,1024); print(r['us'], r['exact_fp32_kernel_us'])"
done
