cd /root/repo
L0=/root/repo/vqvae-vqgan-pytorch-lightning_amd/libvqk.so; L1=/root/repo/ab_libs/libvqk_wl1.so
VQK_LIB=$L1 python -m pytest tests/test_gpu_conv_mx.py tests/test_gpu_conv_gnstats.py tests/test_gpu_conv_ups_phase.py tests/test_gpu_pooled_backward.py tests/test_gpu_conv_s2.py tests/test_gpu_mx_vs_torch.py -q -x 2>&1 | tail -3
for rep in 1 2; do for lib in $L0 $L1; do
  echo "== $lib"; VQK_LIB=$lib VQK_NO_WGRAD=1 python tools/convbench.py bf16 20 2>&1 | grep -E "k3|weighted"
done; done
bash tools/ab_env_multi.sh "VQK_LIB=$L0" "VQK_LIB=$L1"
