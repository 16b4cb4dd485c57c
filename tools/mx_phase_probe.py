#!/usr/bin/env python
"""Where do the matrix waves of conv3x3_mx_kernel spend their cycles?  (attribute the idle matrix pipe.)
Needs the instrumented build:  tools/ab_build.sh probe -DVQK_MXABL=64;  VQK_LIB=ab_libs/libvqk_probe.so python tools/mx_phase_probe.py
Every matrix wave reports (s_memtime, shader clock): cycles in its unit loop, cycles between "my MFMAs are issued" and "the unit
barrier released me" (= waiting for the auxiliary waves: halo DMA landing / drain of the previous tile), cycles parking finished tiles
(accumulators -> bf16 -> LDS), and its unit count.  A unit's MFMAs alone are 144 x 32 = 4608 pipe cycles.
args: cin cout hw [batch]"""
import ctypes
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')
cin, cout, hw = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (128, 128, 256)))
n = int(sys.argv[4]) if len(sys.argv) > 4 else 32
BF, CL = torch.bfloat16, torch.channels_last
lib = native.lib()
x = torch.randn(n, cin, hw, hw, device='cuda').to(BF).contiguous(memory_format=CL)
w = torch.randn(cout, 3, 3, cin, device='cuda') * 0.03
lay = ops.weight_layout(BF, n, hw, hw, cin, cout, 3, False)
wq = ops.pack_weights(w.reshape(-1), BF, cout, cin, 3, False, lay)
y = ops.empty_nhwc(n, cout, hw, hw, BF, x.device)
dbg = torch.zeros(4096 * 8, dtype=torch.int64, device='cuda')
stream = ops._stream()


def launch():
    st = lib.vqk_conv2d_fprop_gnstats(ops.dcode(BF), x.data_ptr(), wq.data_ptr(), 0, 0, y.data_ptr(), n, hw, hw, cin, cout, 3, 0, 0, 1.0,
                                      dbg.data_ptr(), 32, ops.zero_page(x.device).data_ptr(), stream)
    native.check(st, 'conv2d_fprop_gnstats (instrumented)')


for _ in range(3):
    launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
dbg.zero_()
e0.record()
launch()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
d = dbg.view(-1, 8).cpu().double()
d = d[d[:, 3] > 0]
units = d[:, 3]
tot, bar, park = d[:, 0] / units, d[:, 1] / units, d[:, 2] / units
loop = tot - bar - park
flops = 2.0 * n * hw * hw * cin * cout * 9
print(f'{cin}->{cout} @{hw}x{hw}, {n} images: {us:.1f} us = {flops / us / 1e6:.0f} TF (instrumented build: s_memtime costs a few %); '
      f'{len(d)} matrix waves, {units.mean():.1f} units each')
print(f'per unit and matrix wave (shader cycles; the unit\'s 144 MFMAs alone = 4608):')
print(f'  unit loop total     {tot.mean():8.0f}   (min {tot.min():.0f}, max {tot.max():.0f})')
print(f'  MFMA phase section  {loop.mean():8.0f}   = {loop.mean() / 4608:.3f} x the MFMAs alone  (in-loop stalls: LDS fragment reads, weight loads, issue)')
print(f'  barrier wait        {bar.mean():8.0f}   = {100 * bar.mean() / tot.mean():.1f} % of the loop  (auxiliary waves not done: halo DMA / drain)')
print(f'  tile park           {park.mean():8.0f}   = {100 * park.mean() / tot.mean():.1f} % of the loop  (accumulators -> LDS, once per tile)')
nun = d[:, 6]
tiles = units / nun
first, last = d[:, 4] / tiles, d[:, 5] / tiles
mid = (loop * units - d[:, 4] - d[:, 5]) / (units - 2 * tiles).clamp(min=1)
print(f'  MFMA section by the unit\'s place in its tile ({int(nun.mean())} units per tile): first {first.mean():.0f} (the auxiliary waves drain the '
      f'PREVIOUS tile beside it), middle {mid.mean():.0f}, last {last.mean():.0f}')
print(f'  matrix pipe duty if the MFMAs were the only pipe work: {4608 / tot.mean():.3f}')
clock = d[:, 0].mean() / us * 1e-3
print(f'  shader clock seen by s_memtime over the loop: ~{clock:.2f} GHz (loop cycles / kernel time; the kernel also has a prologue)')
