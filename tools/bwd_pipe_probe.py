#!/usr/bin/env python
"""Does a half-batch software pipeline shorten the BACKWARD chain of the large maps?  Per layer the product runs
data-gradient conv (alone) -> [GroupNorm backward || weight gradient on the side stream] -> join: the chain dgrad + GroupNorm is
serial (GroupNorm needs the conv's whole output, the next conv needs GroupNorm's).  GroupNorm is per sample, so two half-batch
chains are exact; started one kernel apart, dgrad(B) runs beside GroupNorm(A).  L layers on [N, C, H, W] bf16:
(a) the product's schedule, (b) two half chains on two streams + whole-batch weight gradients on the side stream, (c) as (b) with
the weight gradients per half at the end of each chain's layer.  ms per chain."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')
BF = torch.bfloat16


def run(n, c, h, w, L=6, reps=5):
    dev = 'cuda'
    g = torch.Generator().manual_seed(0)
    x = ops.nhwc(torch.randn(n, c, h, w, generator=g).to(BF).to(dev))
    dy0 = ops.nhwc(torch.randn(n, c, h, w, generator=g).to(BF).to(dev))
    wm = (torch.randn(c, 3, 3, c, generator=g) * 0.03).to(dev).reshape(-1)
    gw, gb = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    tw, tb = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    tgt = torch.zeros(c, 3, 3, c, device=dev).permute(0, 3, 1, 2)
    groups = 32
    a, st = ops.raw_gn_forward(x, gw, gb, groups, 1e-6, True)
    st2 = st.view(n, -1)
    lay = ops.weight_layout(BF, n, h, w, c, c, 3, False)
    wt = ops.pack_weights(wm, BF, c, c, 3, True, lay)
    layh = ops.weight_layout(BF, n // 2, h, w, c, c, 3, False)
    assert lay == layh
    main, side, pipe = torch.cuda.current_stream(), torch.cuda.Stream(), torch.cuda.Stream()
    bufs = [torch.empty_like(x) for _ in range(4)]       # (timing probe: three rotating outputs + one dgrad buffer)
    cap = ops._wgrad_cap(h * w)
    lib = native.lib()
    hb = n // 2

    def serial():
        lib.vqk_conv_set_block_caps(ops.OVERLAP_STREAM_BLOCKS, cap)
        t = dy0
        for l in range(L):
            d_a = ops.raw_conv_fprop(t, wt, None, None, 3, False, 0, BF, c, lay)
            side.wait_event(ops._fork_point(main))
            with torch.cuda.stream(side):
                ops.raw_conv_wgrad(a, t, 3, False, out=tgt)
            t = ops.raw_gn_backward(x, st, gw, gb, d_a, groups, True, tw, tb)[0]
            main.wait_stream(side)
        lib.vqk_conv_set_block_caps(0, 0)

    def piped(wg_half):
        lib.vqk_conv_set_block_caps(ops.OVERLAP_STREAM_BLOCKS, cap)
        start = ops._fork_point(main)
        pipe.wait_event(start); side.wait_event(start)
        T = dy0
        first = torch.cuda.Event()
        for l in range(L):
            D, Tn = bufs[3], bufs[l % 3]
            evs = []
            for k, (strm, sl) in enumerate(((main, slice(0, hb)), (pipe, slice(hb, n)))):
                with torch.cuda.stream(strm):
                    if l == 0 and k == 1:
                        strm.wait_event(first)             # chain B starts one kernel late
                    ops.raw_conv_fprop(T[sl], wt, None, None, 3, False, 0, BF, c, lay, out=D[sl])
                    if l == 0 and k == 0:
                        first.record(strm)
                    ops.raw_gn_backward(x[sl], st2[sl], gw, gb, D[sl], groups, True, tw, tb, out=Tn[sl])
                    if wg_half:
                        ops.raw_conv_wgrad(a[sl], T[sl], 3, False, out=tgt)
                    evs.append(ops._fork_point(strm))
            if not wg_half:
                with torch.cuda.stream(side):
                    ops.raw_conv_wgrad(a, T, 3, False, out=tgt)    # T of this layer: ready since the previous layer's events
                for e in evs:
                    side.wait_event(e)
            T = Tn
        main.wait_stream(pipe); main.wait_stream(side)
        lib.vqk_conv_set_block_caps(0, 0)

    out = {}
    for name, fn in (('serial', serial), ('piped', lambda: piped(False)), ('piped_wg_half', lambda: piped(True))):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / reps / L * 1e3
    print(f'n{n} c{c} {h}x{w} L={L}: per layer  serial {out["serial"]:.0f} us | piped {out["piped"]:.0f} us | piped, wgrad per half '
          f'{out["piped_wg_half"]:.0f} us', flush=True)


if __name__ == '__main__':
    run(32, 128, 256, 256)
    run(32, 256, 128, 128)
    run(32, 256, 64, 64)
    run(32, 512, 32, 32)
