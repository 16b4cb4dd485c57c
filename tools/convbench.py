#!/usr/bin/env python
"""Per-shape micro-benchmark of the conv kernels (fprop == dgrad kernel, wgrad) on the BASELINE layer shapes
(bs=32).  Usage: python tools/convbench.py [bf16|f32|x3] [iters]   (x3: fp32 storage, split products on the bf16 pipe)"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')

SHAPES = [  # cin, cout, hw_out, k, ups, count(enc+dec fwd)
    (128, 128, 256, 3, 0, 4), (128, 128, 256, 3, 1, 1), (128, 256, 128, 3, 0, 1), (256, 256, 128, 3, 0, 3),
    (128, 128, 128, 3, 0, 4), (128, 128, 128, 3, 1, 1), (256, 256, 64, 3, 0, 4), (256, 256, 64, 3, 1, 1),
    (256, 128, 64, 3, 0, 1), (128, 128, 64, 3, 0, 3), (256, 512, 32, 3, 0, 1), (512, 512, 32, 3, 0, 3),
    (256, 256, 32, 3, 0, 4), (256, 256, 32, 3, 1, 1), (512, 512, 16, 3, 0, 8), (512, 256, 16, 3, 0, 1),
    (256, 256, 16, 3, 0, 3), (256, 512, 16, 3, 0, 1), (128, 256, 128, 1, 0, 1), (8, 128, 256, 3, 0, 1),
    (128, 8, 256, 3, 0, 1),
]


def timeit(fn, iters):
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


def main():
    dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
    if len(sys.argv) > 1 and sys.argv[1] == 'x3':
        ops.set_conv_products('bf16x3')
        if os.environ.get('VQK_X3_WL'):
            native.lib().vqk_set_tuning(b'X3_WL', int(os.environ['VQK_X3_WL']))
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    n = 32
    tot_f = tot_w = 0.0
    variant = int(os.environ.get('VQK_VARIANT', '-1'))
    native.lib().vqk_conv_set_variant(variant)
    skip_w = os.environ.get('VQK_NO_WGRAD') == '1'
    skip_f = os.environ.get('VQK_NO_FPROP') == '1'
    print(f'variant {variant}')
    print(f'{"shape":38s} {"GFLOP":>8s} {"fprop us":>9s} {"TF":>7s} {"wgrad us":>9s} {"TF":>7s}')
    for cin, cout, hw, k, ups, cnt in SHAPES:
        hin = hw >> ups
        x = torch.randn(n, cin, hin, hin, device='cuda').to(dt).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, k, k, cin, device='cuda') * 0.05).to(dt)
        if os.environ.get('VQK_ZERO') == '1':      # DVFS probe: zero operands draw less power, the clock stays up
            x.zero_(); w.zero_()
        dy = torch.randn(n, cout, hw, hw, device='cuda').to(dt).contiguous(memory_format=torch.channels_last)
        fl = 2.0 * n * hw * hw * cin * cout * k * k
        layout = ops.weight_layout(dt, n, hin, hin, cin, cout, k, bool(ups))
        wq = ops.pack_weights(w.float().reshape(-1), dt, cout, cin, k, False, layout)
        if os.environ.get('VQK_GNSTATS') == '1' and k == 3 and layout == 1 and cout % 128 == 0:      # fused GroupNorm sums
            fn = lambda: ops.raw_conv_fprop_gnstats(x, wq, None, None, bool(ups), cout, 32)
            if fn() is None:
                fn = lambda: ops.raw_conv_fprop(x, wq, None, None, k, bool(ups), 0, dt, cout, layout)
        else:
            fn = lambda: ops.raw_conv_fprop(x, wq, None, None, k, bool(ups), 0, dt, cout, layout)
        tf = 1.0 if skip_f else timeit(fn, iters)
        tw = 1.0 if skip_w else timeit(lambda: ops.raw_conv_wgrad(x, dy, k, bool(ups), x3=ops.X3), iters)
        tot_f += tf * cnt
        tot_w += tw * cnt
        print(f'{cin:4d}->{cout:4d} @{hw:3d}^2 k{k} ups{ups} x{cnt:<2d}          {fl / 1e9:8.1f} {tf * 1e6:9.1f} {fl / tf / 1e12:7.1f} '
              f'{tw * 1e6:9.1f} {fl / tw / 1e12:7.1f}')
    print(f'weighted per-step: fprop {tot_f * 1e3:.2f} ms (x2 with dgrad), wgrad {tot_w * 1e3:.2f} ms')


if __name__ == '__main__':
    main()
