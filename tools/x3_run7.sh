cd /root/repo
python -m pytest tests/test_gpu_conv_x3.py -q 2>&1 | tail -8
VQK_NO_FPROP=1 python tools/convbench.py x3 10 2>&1 | tail -23 > gpurun_out/x3_wg_direct.txt
VQK_NO_FPROP=1 VQK_X3_WGRAD_FOLD=1 python tools/convbench.py x3 10 2>&1 | tail -23 > gpurun_out/x3_wg_fold.txt
paste <(cut -c1-40,60-80 gpurun_out/x3_wg_direct.txt) <(cut -c60-80 gpurun_out/x3_wg_fold.txt)
for c in 800 3200; do echo coef $c; VQK_X3_WGRAD_COEF_E4=$c VQK_NO_FPROP=1 python tools/convbench.py x3 10 2>&1 | tail -1; done
