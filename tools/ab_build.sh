#!/bin/bash
# build an A/B variant of libvqk.so with extra -D flags: tools/ab_build.sh <tag> [-DFOO=1 ...]  -> ab_libs/libvqk_<tag>.so
# (ab_libs/ is git-ignored by extension and travels to the GPU box with the snapshot; VQK_LIB=<path> selects it)
# AB_SRC selects the translation unit that gets the flags (default conv_mx.hip; AB_SRC=conv.hip for the other conv kernels)
set -e
tag=$1; shift
src=${AB_SRC:-conv_mx.hip}
obj=${src%.hip}.o
cd "$(dirname "$0")/../vqvae-vqgan-pytorch-lightning_amd/csrc"
mkdir -p ../../ab_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics "$@" -c $src -o ../../ab_libs/${obj%.o}_$tag.o
objs=""
for o in vq.o vq_filter.o entropy.o conv.o conv_wgrad.o conv_mx.o conv_wgmx.o conv_x3.o conv_thin_f32.o conv_edge.o norm.o pointwise.o optim.o stylegan_ops.o gan_ops.o calib.o api.o; do
  if [ "$o" = "$obj" ]; then objs="$objs ../../ab_libs/${obj%.o}_$tag.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab_libs/libvqk_$tag.so $objs
echo ab_libs/libvqk_$tag.so
