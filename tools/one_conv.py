#!/usr/bin/env python
"""Run ONE conv shape repeatedly (for rocprofv3 --pmc runs).  args: cin cout hw k ups which(fprop|wgrad) iters
VQK_ONE_CONV_MODE=x3: fp32 tensors, split products on the bf16 pipe (csrc/conv_x3.hip); f32: the exact-fp32 kernels"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
cin, cout, hw, k, ups = (int(a) for a in sys.argv[1:6])
which, iters = sys.argv[6], int(sys.argv[7])
mode = os.environ.get('VQK_ONE_CONV_MODE', 'bf16')
dt = torch.bfloat16 if mode == 'bf16' else torch.float32
if mode == 'x3':
    ops.set_conv_products('bf16x3')
n, hin = 32, hw >> ups
x = torch.randn(n, cin, hin, hin, device='cuda').to(dt).contiguous(memory_format=torch.channels_last)
w = (torch.randn(cout, k, k, cin, device='cuda') * 0.05).to(dt)
dy = torch.randn(n, cout, hw, hw, device='cuda').to(dt).contiguous(memory_format=torch.channels_last)
layout = ops.weight_layout(dt, n, hin, hin, cin, cout, k, bool(ups))
wq = ops.pack_weights(w.float().reshape(-1), dt, cout, cin, k, False, layout)
for _ in range(iters):
    if which == 'fprop':
        ops.raw_conv_fprop(x, wq, None, None, k, bool(ups), 0, dt, cout, layout)
    else:
        ops.raw_conv_wgrad(x, dy, k, bool(ups), x3=ops.X3)
torch.cuda.synchronize()
