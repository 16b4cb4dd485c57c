#!/usr/bin/env python
"""Stride-2 3x3 conv of the discriminator (blurred (2h+1)^2 input -> h^2) and its data gradient: matrix/auxiliary-wave form
(vqk_conv2d_s2_fprop / vqk_conv2d_s2_dgrad) against the im2col kernel (tuning slot MX_S2 = 0), hipEvent-timed.
usage: PYTHONPATH=. python tools/s2_bench.py [batch] [iters]"""
import importlib
import sys

import torch

ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
_native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')
DEV, BF, CL = 'cuda:0', torch.bfloat16, torch.channels_last
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
SHAPES = [(128, 256, 128), (256, 512, 64), (512, 512, 32), (512, 512, 16)]      # (cin, cout, h_out) of b256 .. b32


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


lib = _native.lib()
print(f'{"shape":34s} {"fprop mx":>9s} {"im2col":>9s} {"TF mx":>7s} | {"dgrad mx":>9s} {"im2col":>9s} {"TF mx":>7s}   (us)')
for cin, cout, ho in SHAPES:
    x = torch.randn(n, cin, 2 * ho + 1, 2 * ho + 1, device=DEV).to(BF).contiguous(memory_format=CL)
    w = torch.randn(cout, cin, 3, 3, device=DEV).contiguous(memory_format=CL)
    dy = torch.randn(n, cout, ho, ho, device=DEV).to(BF).contiguous(memory_format=CL)
    flops = 2.0 * n * ho * ho * cin * cout * 9
    res = []
    for on in (1, 0):
        lib.vqk_set_tuning(b'MX_S2', on)
        with torch.no_grad():
            tf = timed(lambda: ops.ConvActFn.apply(x, w, None, 3, 2, 0, 0, 1.0, 1.0, None))
            tb = timed(lambda: ops.ConvDgradFn.apply(dy, w, 3, 2, 0, 1.0, cin, cout, 2 * ho + 1, 2 * ho + 1))
        res.append((tf, tb))
    lib.vqk_set_tuning(b'MX_S2', 1)
    (f1, b1), (f0, b0) = res
    print(f'n{n} {cin:3d}->{cout:3d} @{2 * ho + 1:3d}^2 -> {ho:3d}^2      {f1:9.1f} {f0:9.1f} {flops / f1 / 1e6:7.0f} | {b1:9.1f} {b0:9.1f} {flops / b1 / 1e6:7.0f}')
