#!/usr/bin/env python
"""Does running GroupNorm passes per batch chunk (working set inside the 256 MB Infinity Cache) beat whole-batch passes?"""
import importlib, os, sys
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


for c, hw in [(128, 256), (256, 128), (128, 128)]:
    x = torch.randn(32, c, hw, hw, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
    dy = torch.randn_like(x)
    w = torch.ones(c, device='cuda'); b = torch.zeros(c, device='cuda')
    stats = ops.raw_gn_stats(x, 32, 1e-6)
    dw = torch.zeros(c, device='cuda'); db = torch.zeros(c, device='cuda')
    for chunk in (32, 16, 8, 4, 2):
        def fwd():
            for i in range(0, 32, chunk):
                ops.raw_gn_forward(x[i:i + chunk], w, b, 32, 1e-6, True)
        def bwd():
            for i in range(0, 32, chunk):
                ops.raw_gn_backward(x[i:i + chunk], stats[i * 64:(i + chunk) * 64], w, b, dy[i:i + chunk], 32, True, dw, db)
        print(f'C={c} {hw}^2 chunk {chunk:2d}: fwd {timeit(fwd) * 1e6:7.1f} us  bwd {timeit(bwd) * 1e6:7.1f} us')
