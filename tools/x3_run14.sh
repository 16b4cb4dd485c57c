cd /root/repo
python -m pytest tests/test_gpu_conv_mx.py tests/test_gpu_conv_gnstats.py tests/test_gpu_conv_ups_phase.py tests/test_gpu_pooled_backward.py tests/test_gpu_conv_s2.py tests/test_gpu_tile_queue.py tests/test_gpu_mx_vs_torch.py -q -x 2>&1 | tail -3
for sh in "128 128 256" "256 256 128"; do VQK_LIB=/root/repo/ab_libs/libvqk_probe.so python tools/mx_phase_probe.py $sh; done 2>&1 | grep -E "TF|first|duty"
for rep in 1 2; do
for lib in /root/repo/ab_libs/libvqk_nopace.so /root/repo/vqvae-vqgan-pytorch-lightning_amd/libvqk.so; do
  echo "== $lib"; VQK_LIB=$lib VQK_NO_WGRAD=1 python tools/convbench.py bf16 20 2>&1 | grep -E "@256|@128|weighted"
done; done
bash tools/ab_env_multi.sh "VQK_LIB=/root/repo/ab_libs/libvqk_nopace.so" "VQK_LIB=/root/repo/vqvae-vqgan-pytorch-lightning_amd/libvqk.so"
