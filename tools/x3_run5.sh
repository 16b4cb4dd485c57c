cd /root/repo
python -m pytest tests/test_gpu_gan.py -q -s -k "optimizer_overlap" 2>&1 | grep -E "GAN optimizer|Error|assert|^E |passed|failed" | head -30
