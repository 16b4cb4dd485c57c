#!/usr/bin/env python
"""GroupNorm kernel micro-benchmark at the BASELINE activation shapes (bs=32, bf16): GB/s per kernel."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for c, hw in [(128, 256), (256, 128), (128, 128), (256, 64), (512, 32), (512, 16)]:
    x = torch.randn(32, c, hw, hw, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
    dy = torch.randn_like(x)
    w = torch.ones(c, device='cuda')
    b = torch.zeros(c, device='cuda')
    stats = ops.raw_gn_stats(x, 32, 1e-6)
    nbytes = x.numel() * 2
    t1 = timeit(lambda: ops.raw_gn_stats(x, 32, 1e-6))
    t2 = timeit(lambda: ops.raw_gn_apply(x, stats, w, b, 32, True))
    t12 = timeit(lambda: ops.raw_gn_forward(x, w, b, 32, 1e-6, True))
    t3 = timeit(lambda: ops.raw_gn_backward(x, stats, w, b, dy, 32, True))
    print(f'C={c:3d} {hw:3d}^2  stats {t1 * 1e6:7.1f} us {nbytes / t1 / 1e12:5.2f} TB/s | apply {t2 * 1e6:7.1f} us '
          f'{2 * nbytes / t2 / 1e12:5.2f} TB/s | fwd(2 kernels) {t12 * 1e6:7.1f} us | bwd(2 kernels) {t3 * 1e6:7.1f} us {5 * nbytes / t3 / 1e12:5.2f} TB/s')
