"""Batch slices as kernel operands (``out=`` of ops.raw_conv_fprop / ops.raw_gn_backward): GroupNorm is per sample
(vqvae/modules/autoencoder.py:25-39) and a conv output pixel depends on its own sample only, so a launch on x[a:b] that writes
y[a:b] of a larger nhwc tensor must give exactly the whole-batch result (tools/bwd_pipe_probe.py builds its half-batch chains on this)."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu

ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
DEV, BF, CL = 'cuda:0', torch.bfloat16, torch.channels_last


def test_conv_and_group_norm_backward_on_batch_halves_equal_the_whole_batch():
    g = torch.Generator(device=DEV).manual_seed(3)
    n, c, h = 6, 128, 64
    x = torch.randn(n, c, h, h, device=DEV, generator=g).to(BF).contiguous(memory_format=CL)
    dy = torch.randn(n, c, h, h, device=DEV, generator=g).to(BF).contiguous(memory_format=CL)
    wm = (torch.randn(c, 3, 3, c, device=DEV, generator=g) / 34).reshape(-1)
    gw, gb = torch.rand(c, device=DEV, generator=g) + 0.5, torch.randn(c, device=DEV, generator=g) * 0.1
    lay = ops.weight_layout(BF, n, h, h, c, c, 3, False)
    assert lay == ops.weight_layout(BF, n // 2, h, h, c, c, 3, False)      # (the kernel choice does not depend on the batch size)
    wq = ops.pack_weights(wm, BF, c, c, 3, False, lay)
    want = ops.raw_conv_fprop(dy, wq, None, None, 3, False, 0, BF, c, lay)
    got = torch.empty_like(want)
    for sl in (slice(0, n // 2), slice(n // 2, n)):
        r = ops.raw_conv_fprop(dy[sl], wq, None, None, 3, False, 0, BF, c, lay, out=got[sl])
        assert r.data_ptr() == got[sl].data_ptr()
    assert torch.equal(got, want)
    _, st = ops.raw_gn_forward(x, gw, gb, 32, 1e-6, True)
    dw0, db0 = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
    dx0, _, _ = ops.raw_gn_backward(x, st, gw, gb, dy, 32, True, dw0, db0)
    dw1, db1 = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
    dx1 = torch.empty_like(dx0)
    st2 = st.view(n, -1)
    for sl in (slice(0, n // 2), slice(n // 2, n)):
        ops.raw_gn_backward(x[sl], st2[sl], gw, gb, dy[sl], 32, True, dw1, db1, out=dx1[sl])
    # (group sums are fp64 atomics over a block partition that depends on the launch's sample count: a few outputs round the other way)
    assert float((dx1 != dx0).float().mean()) < 1e-3 and float((dx1.float() - dx0.float()).norm() / dx0.float().norm()) < 1e-4
    assert float((dw1 - dw0).norm() / dw0.norm()) < 1e-5 and float((db1 - db0).norm() / db0.norm()) < 1e-5    # fp32 sums, other order
