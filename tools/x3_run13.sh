cd /root/repo
for sh in "128 128 256" "256 256 128" "256 256 64" "512 512 32"; do VQK_LIB=/root/repo/ab_libs/libvqk_probe.so python tools/mx_phase_probe.py $sh; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/round6_mx_phase_attribution.txt
