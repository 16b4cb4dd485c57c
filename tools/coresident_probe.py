#!/usr/bin/env python
"""Weight gradient (MFMA-bound) beside the GroupNorm backward (HBM-bound) of the same ResBlock, as in ops.ResBlockFn.backward:
time of each alone, of the pair back to back, and of the pair on two streams, per grid cap of the weight gradient.  The
weight-gradient block (8 waves x 256 registers, 120 KiB LDS) takes a CU's whole register file: nothing runs BESIDE it on
that CU, the two kernels only share the chip CU by CU.  (Round 4 also built a 384-thread form with two auxiliary waves, which
leaves half of two SIMDs' registers to GroupNorm waves: 4.5x slower -- 2609 against 586 us at 128 ch @256^2 -- because twenty
LDS-DMA pieces per auxiliary wave and patch do not issue in a patch's MFMA time; removed, profiles/round4_coresident_probe.txt.)
Usage: python tools/coresident_probe.py [cap ...]   (cap = grid cap of the weight gradient; default 256 320 512)"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')
lib = native.lib()


def timeit(fn, reps=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def run(n, c, h, caps):
    dev = 'cuda'
    g = torch.Generator().manual_seed(0)
    mk = lambda: ops.nhwc(torch.randn(n, c, h, h, generator=g).to(torch.bfloat16).to(dev))
    a, dy, x, dg = mk(), mk(), mk(), mk()
    gw, gb = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    _, stats = ops.raw_gn_forward(x, gw, gb, 32, 1e-6, True)
    dw = torch.zeros(c, 3, 3, c, device=dev).permute(0, 3, 1, 2)
    dgw, dgb = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    side = torch.cuda.Stream()
    wgrad = lambda: ops.raw_conv_wgrad(a, dy, 3, False, out=dw)
    gnb = lambda: ops.raw_gn_backward(x, stats, gw, gb, dg, 32, True, dgw, dgb)

    def pair(cap):
        main = torch.cuda.current_stream()
        lib.vqk_conv_set_block_caps(512, cap)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            wgrad()                                   # launched first, as in the step
        gnb()
        main.wait_stream(side)
        lib.vqk_conv_set_block_caps(0, 0)

    for _ in (0,):
        tw, tg = timeit(wgrad), timeit(gnb)
        line = f'{c}@{h}x{h} bs{n}: wgrad {tw:7.1f} us  gn_bwd {tg:7.1f} us  serial {tw + tg:7.1f}'
        for cap in caps:
            lib.vqk_conv_set_block_caps(512, cap)
            twc = timeit(wgrad)
            lib.vqk_conv_set_block_caps(0, 0)
            line += f' | cap {cap}: wgrad alone {twc:7.1f}, pair {timeit(lambda: pair(cap)):7.1f}'
        print(line, flush=True)


if __name__ == '__main__':
    caps = [int(v) for v in sys.argv[1:]] or [256, 320, 512]
    for n, c, h in ((32, 128, 256), (32, 256, 128), (32, 128, 128), (32, 256, 64), (32, 512, 32)):
        run(n, c, h, caps)
