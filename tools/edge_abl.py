#!/usr/bin/env python
"""time the two edge weight-gradient launches at 256x256, bs 32 (VQK_LIB selects an ablation build)"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
dt, cl = torch.bfloat16, torch.channels_last
for cin, cout in ((8, 128), (128, 8)):
    x = torch.randn(32, cin, 256, 256, device='cuda').to(dt).contiguous(memory_format=cl)
    dy = torch.randn(32, cout, 256, 256, device='cuda').to(dt).contiguous(memory_format=cl)
    out = torch.zeros((cout, 3, 3, cin), dtype=torch.float32, device='cuda').permute(0, 3, 1, 2)
    for _ in range(3):
        ops.raw_conv_wgrad(x, dy, 3, False, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.raw_conv_wgrad(x, dy, 3, False, out=out)
    e1.record(); torch.cuda.synchronize()
    print(f'{cin}->{cout}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us', end='   ')
print()
