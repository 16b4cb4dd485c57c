#!/usr/bin/env python
"""Does a half-batch software pipeline hide the GroupNorm-apply passes of the forward under the convs?  Chain of L
(GroupNorm-apply + SiLU -> 3x3 conv) layers on [N, C, H, W] bf16, (a) whole batch on one stream, (b) two half batches on two
streams, the second started one kernel late so that conv(A) runs beside gn(B).  GroupNorm is per sample, so the halves are
exact.  Prints ms per chain for both."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')


def run(n, c, h, w, L=8, reps=6, cap=0):
    dev = 'cuda'
    g = torch.Generator().manual_seed(0)
    x = ops.nhwc((torch.randn(n, c, h, w, generator=g)).to(torch.bfloat16).to(dev))
    wm = (torch.randn(c, 3, 3, c, generator=g) * 0.03).to(dev).reshape(-1)
    gw, gb = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    groups = 32

    def chain(xx, nloc):
        lay = ops.weight_layout(torch.bfloat16, nloc, h, w, c, c, 3, False)
        wq = ops.pack_weights(wm, torch.bfloat16, c, c, 3, False, lay)
        stats = ops.raw_gn_stats(xx, groups, 1e-6)
        def f(first_evt=None, wait_evt=None):
            t = xx
            for l in range(L):
                if l == 0 and wait_evt is not None:
                    torch.cuda.current_stream().wait_event(wait_evt)
                y = ops.raw_gn_apply(t, stats, gw, gb, groups, True)
                if l == 0 and first_evt is not None:
                    first_evt.record()
                t = ops.raw_conv_fprop(y, wq, None, None, 3, False, 0, torch.bfloat16, c, lay)
            return t
        return f

    full = chain(x, n)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    xa, xb = x[: n // 2], x[n // 2:]
    with torch.cuda.stream(s1):
        fa = chain(xa, n // 2)
    with torch.cuda.stream(s2):
        fb = chain(xb, n // 2)
    torch.cuda.synchronize()

    def t_serial():
        full()

    def t_pipe():
        native.lib().vqk_conv_set_block_caps(cap, cap)        # persistent conv grid capped: room for the GroupNorm blocks
        e = torch.cuda.Event()
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            fa(first_evt=e)
        with torch.cuda.stream(s2):
            fb(wait_evt=e)
        cur.wait_stream(s1); cur.wait_stream(s2)
        native.lib().vqk_conv_set_block_caps(0, 0)

    out = {}
    for name, fn in (('serial', t_serial), ('pipelined', t_pipe)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / reps
    print(f'n{n} c{c} {h}x{w} L={L} cap={cap}: serial {out["serial"]:.3f} ms, pipelined halves {out["pipelined"]:.3f} ms '
          f'({100 * (1 - out["pipelined"] / out["serial"]):+.1f} % saved)')


if __name__ == '__main__':
    for cap in (0, 448, 384, 320, 256):
        run(32, 128, 256, 256, cap=cap)
    for cap in (0, 384, 320):
        run(32, 256, 128, 128, cap=cap)
