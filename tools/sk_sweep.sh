#!/bin/bash
# sweep of the general conv kernel's split-K knobs on the VQ-GAN step's layer list (tools/gan_convs.py): total us of the
# im2col-kernel layers (lay0) per setting
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "1024 4 16" "512 4 16" "1024 8 16" "1024 4 8" "2048 4 32" "1024 2 16" "0 4 16"; do
  set -- $cfg
  VQK_SK_BLOCKS=$1 VQK_SK_MINSTEPS=$2 VQK_SK_MAXMB=$3 python $R/tools/gan_convs.py > /tmp/gc.txt 2>&1 < /dev/null
  echo "blocks=$1 minsteps=$2 maxmb=$3: lay0 total $(grep lay0 /tmp/gc.txt | awk '{s+=$1} END {print s}') us"
  grep -E "mode2 -> (9x9|17x17|33x33|65x65)|n16 8x8 cin512 cout512 k3 s1 pad1|17x17 cin512 cout512 k3 s2|33x33 cin512 cout512 k3 s2| 1x1 cin8192" /tmp/gc.txt | awk '{printf "    %s %s %s %s %s %s\n", $1, $3, $6, $7, $8, $14}'
done
