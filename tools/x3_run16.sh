cd /root/repo
python -m pytest tests/test_gpu_conv_x3.py -q 2>&1 | tail -2
VQK_NO_FPROP=1 python tools/convbench.py x3 10 2>&1 | grep -E "k3|weighted" | head -19
cd /tmp && export TMPDIR=/tmp VQK_ONE_CONV_MODE=x3
rm -rf /tmp/pmc_x3; rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_x3 -o r -- python /root/repo/tools/one_conv.py 128 128 256 3 0 wgrad 6 > /tmp/pmc_x3.log 2>&1
python /root/repo/tools/pmc_summary.py $(find /tmp/pmc_x3 -name '*.db' | head -1) x3_kernel
