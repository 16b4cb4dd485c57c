cd /root/repo
for wl in 1 0 1 0; do echo "X3_WL=$wl"; VQK_X3_WL=$wl VQK_NO_WGRAD=1 python tools/convbench.py x3 10 2>&1 | grep -E "weighted|128-> 128 @256\^2 k3 ups0|256-> 256 @128|512-> 512 @ 16"; done
