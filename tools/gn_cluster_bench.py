#!/usr/bin/env python
"""GroupNorm backward on the mid-size maps (bs 32, bf16): single-kernel cluster form (vqk_gn_backward_ws) against the
two-kernel form (tuning slot GN_CLUSTER_MAX_HW = 0), same box, us per call and TB/s of algorithmic bytes (3 / 5 passes)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gnbench import timeit
lib = native.lib()
shapes = [(256, 64), (128, 64), (256, 32), (512, 32), (128, 128), (256, 128)] + ([(128, 256)] if '--big' in sys.argv else [])
for c, hw in shapes:
    x = torch.randn(32, c, hw, hw, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
    dy = torch.randn_like(x)
    add = torch.randn_like(x)
    w = torch.ones(c, device='cuda'); b = torch.zeros(c, device='cuda')
    _, stats = ops.raw_gn_forward(x, w, b, 32, 1e-6, True)
    nb = x.numel() * 2
    out = f'C={c:3d} {hw:3d}^2'
    for label, mx in (('cluster', 1 << 30), ('two-kernel', 0)):
        lib.vqk_set_tuning(b'GN_CLUSTER_MAX_HW', mx)
        t = timeit(lambda: ops.raw_gn_backward(x, stats, w, b, dy, 32, True), 20)
        ta = timeit(lambda: ops.raw_gn_backward(x, stats, w, b, dy, 32, True, add=add), 20)
        out += f' | {label}: {t * 1e6:7.1f} us, with skip addend {ta * 1e6:7.1f} us'
    lib.vqk_reset_tuning()
    print(out, flush=True)
