#!/usr/bin/env python
"""GroupNorm backward (reduce pass + apply pass, both read x and dy) on image chunks small enough for the 256-MiB Infinity
Cache: does the apply pass of a chunk find x / dy on die?  GroupNorm is per sample, so chunking is exact.  Times
vqk_gn_backward on [N, C, H, W] bf16 whole and in chunks of 1 / 2 / 4 / 8 images (same stream, back to back)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')


def run(n, c, h, w, groups=32, reps=10, with_add=False):
    dev = 'cuda'
    g = torch.Generator().manual_seed(0)
    x = ops.nhwc(torch.randn(n, c, h, w, generator=g).to(torch.bfloat16).to(dev))
    dy = ops.nhwc(torch.randn(n, c, h, w, generator=g).to(torch.bfloat16).to(dev))
    add = ops.nhwc(torch.randn(n, c, h, w, generator=g).to(torch.bfloat16).to(dev)) if with_add else None
    gw, gb = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    stats = ops.raw_gn_stats(x, groups, 1e-6)
    dw, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    res = {}
    ref = None
    for chunk in (n, 8, 4, 2, 1):
        def fn():
            outs = []
            for i in range(0, n, chunk):
                outs.append(ops.raw_gn_backward(x[i:i + chunk], stats[i * groups * 2:(i + chunk) * groups * 2], gw, gb, dy[i:i + chunk],
                                                groups, True, dw, db, None if add is None else add[i:i + chunk]))
            return outs
        for _ in range(2):
            o = fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()                              # replayed from a hipGraph: no host launch cost in the timing
        with torch.cuda.graph(gr):
            o = fn()
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        res[chunk] = e0.elapsed_time(e1) / reps * 1e3
        full = torch.cat([t[0] if isinstance(t, tuple) else t for t in o], 0)
        if ref is None:
            ref = full
        else:
            assert (ref.float() - full.float()).abs().max().item() <= 0.05 * ref.float().abs().max().item(), 'chunked result differs'
    print(f'gn_bwd n{n} c{c} {h}x{w} add={with_add}: ' + ', '.join(f'chunk {k}: {v:.1f} us' for k, v in res.items()))


if __name__ == '__main__':
    run(32, 128, 256, 256)
    run(32, 128, 256, 256, with_add=True)
    run(32, 256, 128, 128)
    run(32, 128, 128, 128)
    run(32, 256, 64, 64)
