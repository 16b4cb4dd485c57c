R=${GRAFT_REPO_ROOT:-/root/repo}
for t in gd2 gd4 gd5; do echo $t; VQK_LIB=$R/scratch/libvqk_$t.so VQK_GN_CLUSTER_MAX_HW=0 python $R/tools/gnbench.py 2>&1 | tail -6 | cut -c1-170; done
run() { ms=$(env "$@" python $R/bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-events --no-other-configs --traffic off --sustain-s 0 2>/dev/null < /dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'); echo "$* $ms"; }
for rep in 1 2; do
run VQK_LIB=$R/scratch/libvqk_sl64.so VQK_GN_CLUSTER_MAX_HW=0 VQK_VQ_FUSED=0
run A=1
run VQK_LIB=$R/scratch/libvqk_gd2.so
run VQK_LIB=$R/scratch/libvqk_gd4.so
run VQK_LIB=$R/scratch/libvqk_gd5.so
done
