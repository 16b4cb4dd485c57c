#!/usr/bin/env python
"""Per-shape A/B of the persistent conv kernel's tile assignment (tuning slot TILE_QUEUE: 0 static share, 1 first tile static,
2 every tile from the queue): back-to-back launches alone, and while a stand-in for a collective's kernel holds `B` CUs on
another stream (vqk_probe_stream_add: persistent 256-thread blocks with 32 KiB of LDS each).
Usage: python tools/tile_queue_bench.py [iters]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')

SHAPES = [  # cin, cout, hw, k, count per step (fprop + dgrad)
    (128, 128, 256, 3, 8), (256, 256, 128, 3, 6), (128, 128, 128, 3, 8), (256, 256, 64, 3, 8), (128, 128, 64, 3, 6),
    (512, 512, 32, 3, 6), (256, 256, 32, 3, 8), (512, 512, 16, 3, 16), (128, 256, 128, 1, 2),
]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    lib = native.lib()
    n, dt = 32, torch.bfloat16
    side = torch.cuda.Stream()
    src = torch.zeros(64 << 20, device='cuda')
    dst = torch.zeros(64 << 20, device='cuda')
    tot = {}
    print(f'{"shape":26s} {"tiles":>6s} | ' + ' | '.join(f'held {b:3d}: static  first  queue' for b in (0, 16, 32, 64)) + '   (us per launch)')
    for cin, cout, hw, k, cnt in SHAPES:
        x = torch.randn(n, cin, hw, hw, device='cuda').to(dt).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, k, k, cin, device='cuda') * 0.05)
        layout = ops.weight_layout(dt, n, hw, hw, cin, cout, k, False)
        wq = ops.pack_weights(w.reshape(-1), dt, cout, cin, k, False, layout)
        fn = lambda: ops.raw_conv_fprop(x, wq, None, None, k, False, 0, dt, cout, layout)
        best = {}
        for rnd in range(3):                                      # the order of the modes rotates: no mode is always measured first
            for held in (0, 16, 32, 64):
                for mode in [(rnd + i) % 3 for i in range(3)]:
                    lib.vqk_set_tuning(b'TILE_QUEUE', mode)
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize()
                    if held:
                        with torch.cuda.stream(side):     # long enough to cover the timed launches: ~ 40 ms
                            lib.vqk_probe_stream_add(src.data_ptr(), dst.data_ptr(), src.numel() * 4, held, 8, 24, side.cuda_stream)
                        torch.cuda._sleep(2_000_000)      # the stand-in is resident before the first conv block looks for a CU
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(iters):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    t = e0.elapsed_time(e1) / iters * 1e3
                    best.setdefault((held, mode), []).append(t)
        row = []
        for held in (0, 16, 32, 64):
            for mode in (0, 1, 2):
                t = sorted(best[(held, mode)])[1]                 # median of the three rounds
                row.append(t)
                tot[(held, mode)] = tot.get((held, mode), 0.0) + t * cnt
        tiles = n * hw * hw // 256 * (cout // 128)
        print(f'{cin:4d}->{cout:4d} @{hw:3d}^2 k{k} x{cnt:<2d}  {tiles:6d} | ' +
              ' | '.join('          ' + ' '.join(f'{v:6.1f}' for v in row[3 * i:3 * i + 3]) for i in range(4)))
    print('weighted per step (ms):   ' + ' | '.join(f'held {h:3d}: ' + ' '.join(f'{tot[(h, m)] / 1e3:6.2f}' for m in (0, 1, 2)) for h in (0, 16, 32, 64)))
    lib.vqk_reset_tuning()


if __name__ == '__main__':
    main()
