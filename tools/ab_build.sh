#!/bin/bash
# build an A/B variant of libvqk.so with extra -D flags for conv.hip: tools/ab_build.sh <tag> [-DFOO=1 ...]  -> scratch/libvqk_<tag>.so
set -e
tag=$1; shift
cd "$(dirname "$0")/../vqvae-vqgan-pytorch-lightning_amd/csrc"
mkdir -p ../../scratch
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics "$@" -c conv.hip -o ../../scratch/conv_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/libvqk_$tag.so vq.o entropy.o ../../scratch/conv_$tag.o norm.o pointwise.o optim.o stylegan_ops.o gan_ops.o api.o
echo scratch/libvqk_$tag.so
