"""Backward of a ResBlock whose output went through the fused 2x2 average pool (autoencoder.py:63-77 + :89-91) with the
gradient kept at HALF resolution -- conv2's data gradient through the nearest-x2 halo addressing, conv2's weight gradient
and norm1's skip addend from the pooled pixel of each 2x2 block -- against the path that unpools the gradient first
(which the golden / full-size tests pin to the reference): every gradient, same bf16 noise."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu

ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
ae = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.modules.autoencoder')
optim = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.optim')
DEV, BF, CL = 'cuda:0', torch.bfloat16, torch.channels_last


@pytest.mark.parametrize('n,c,h,w', [(2, 128, 64, 64), (3, 256, 48, 64), (1, 128, 64, 128)])
def test_pooled_gradient_path_matches_unpool_path(n, c, h, w):
    torch.manual_seed(c + h)
    blk = ae.ResBlock(c).to(DEV)
    with torch.no_grad():
        blk.norm1.weight.normal_(1.0, 0.2); blk.norm1.bias.normal_(0.0, 0.2)
        blk.norm2.weight.normal_(1.0, 0.2); blk.norm2.bias.normal_(0.0, 0.2)
    opt = optim.FlatAdamW(blk.parameters(), lr=1e-4, betas=(0.0, 0.99), weight_decay=0.0)
    x0 = torch.randn(n, c, h, w, device=DEV).to(BF).contiguous(memory_format=CL)
    dy = torch.randn(n, c, h // 2, w // 2, device=DEV).to(BF).contiguous(memory_format=CL)
    res = {}
    saved, real_unpool, calls = ops.POOLED_BWD, ops.raw_unpool, {False: 0, True: 0}
    try:
        for mode in (False, True):
            ops.POOLED_BWD = mode

            def counting(*a, _m=mode, **k):
                calls[_m] += 1
                return real_unpool(*a, **k)
            ops.raw_unpool = counting
            opt.zero_grad()
            x = x0.clone().requires_grad_(True)
            y = blk(x, pool=True)
            y.backward(dy)
            torch.cuda.synchronize()
            res[mode] = (y.detach().float().clone(), x.grad.float().clone(), opt.flat_g.clone())
    finally:
        ops.POOLED_BWD, ops.raw_unpool = saved, real_unpool
    assert calls == {False: 1, True: 0}, calls            # the pooled path really ran (and never unpooled)
    (y0, dx0, g0), (y1, dx1, g1) = res[False], res[True]
    assert torch.equal(y0, y1)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    assert rel(dx1, dx0) < 6e-3, rel(dx1, dx0)           # one more / one less bf16 rounding of the unpooled gradient
    assert rel(g1, g0) < 6e-3, rel(g1, g0)
    for name, p in blk.named_parameters():
        assert rel(p.grad.float(), p.grad.float()) == 0.0 and float(p.grad.abs().max()) > 0, name


@pytest.mark.parametrize('n,c,h', [(2, 128, 64), (3, 256, 32), (2, 128, 16)])
def test_pooled_data_gradient_in_phase_form(n, c, h):
    """round 5: the data gradient of conv2 + fused average pool from the pooled gradient as ONE forward-type phase launch
    (vqk_conv2d_pooled_dgrad_phase: reversed phase blocks of the conv's phase-form data-gradient operand, 4/9 of the multiply-adds)
    against fp32 PyTorch: d/dx of avg_pool2d(conv2d(x, W)) for the pooled cotangent"""
    g = torch.Generator(device=DEV).manual_seed(n + c + h)
    wgt = torch.nn.Parameter((torch.randn(c, c, 3, 3, device=DEV, generator=g) / (3 * c ** 0.5)).to(torch.bfloat16).float()
                             .contiguous(memory_format=torch.channels_last))
    dyp = torch.randn(n, c, h, h, device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    got = ops.raw_conv_pooled_dgrad_phase(dyp, wgt, 0.25)
    assert got is not None and got.shape == (n, c, 2 * h, 2 * h)
    x = torch.zeros(n, c, 2 * h, 2 * h, device=DEV, requires_grad=True)
    torch.nn.functional.avg_pool2d(torch.nn.functional.conv2d(x, wgt.detach().float(), padding=1), 2).backward(dyp.float())
    err = float((got.float() - x.grad).norm() / x.grad.norm())
    assert err < 6e-3, err                                      # bf16 output + weights pre-summed in fp32 and rounded once
    # and against the tap form it replaces (same operands, bf16 output): agreement to the output rounding
    lay = ops.weight_layout(torch.bfloat16, n, h, h, c, c, 3, True)
    wt = ops.packed_weight(wgt, c, c, torch.bfloat16, 3, True, lay)
    tap = ops._conv_general_raw(dyp, wt, None, None, c, 3, 1, 1, 1, 2 * h, 2 * h, 0, 0.25, 1.0, torch.bfloat16, lay)
    assert float((got.float() - tap.float()).norm() / tap.float().norm()) < 8e-3


@pytest.mark.parametrize('n,c,h,res,gn', [(2, 128, 64, 1, 0), (3, 256, 32, 0, 0), (2, 128, 128, 1, 32), (2, 256, 64, 1, 32)])
def test_pooled_forward_as_4x4_stride2_phase_launch(n, c, h, res, gn):
    """round 5: avg_pool2d(conv2d(x, W) + skip) = the 4x4 stride-2 conv of x + pooled skip, as ONE data-gradient-type phase launch
    with the conv's forward phase operand, phase blocks reversed (vqk_conv2d_pooled_fprop_phase) -- against fp32 PyTorch, against the
    conv + pooling-drain kernel it replaces, and with the GroupNorm sums of the result left for the consumer"""
    g = torch.Generator(device=DEV).manual_seed(n + c + h + res)
    wgt = torch.nn.Parameter((torch.randn(c, c, 3, 3, device=DEV, generator=g) / (3 * c ** 0.5)).to(torch.bfloat16).float()
                             .contiguous(memory_format=torch.channels_last))
    x = torch.randn(n, c, h, h, device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    skip = torch.randn(n, c, h, h, device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if res else None
    skip_p = ops.raw_pool(skip, 0.25) if res else None
    got = ops.raw_conv_pooled_fprop_phase(x, wgt, skip_p, 0.25, gn)
    assert got is not None and got.shape == (n, c, h // 2, h // 2)
    want = torch.nn.functional.conv2d(x.float(), wgt.detach().float(), padding=1)
    if res:
        want = want + skip.float()
    want = torch.nn.functional.avg_pool2d(want, 2)
    assert float((got.float() - want).norm() / want.norm()) < 6e-3
    wq = ops.packed_weight(wgt, c, c, torch.bfloat16, 3, False, 1)
    old = ops.raw_conv_fprop_pooled(x, wq, None, skip, 3, False, c, 0.25)
    assert float((got.float() - old.float()).norm() / old.float().norm()) < 8e-3
    if gn and (h // 2) * (h // 2) > 1024:                       # (maps of <= 1024 pixels: the single-kernel GroupNorm, no hand-off)
        # the sums left in the workspace are those of the STORED tensor: GroupNorm with them == GroupNorm with its own statistics pass
        gw, gb = torch.randn(c, device=DEV, generator=g), torch.randn(c, device=DEV, generator=g)
        assert ops.pending_gn() is not None
        y1, st1 = ops.raw_gn_forward(got, gw, gb, gn, 1e-6, True)              # claims the hand-off
        assert ops.pending_gn() is None
        y2, st2 = ops.raw_gn_forward(got.clone(memory_format=torch.preserve_format), gw, gb, gn, 1e-6, True)
        assert float((st1 - st2).abs().max() / st2.abs().max()) < 1e-4
        assert float((y1.float() - y2.float()).norm() / y2.float().norm()) < 2e-3


@pytest.mark.parametrize('n,cin,cout,h', [(2, 128, 128, 64), (3, 256, 256, 32), (2, 128, 256, 32), (1, 64, 128, 48)])
def test_pooled_weight_gradient_in_phase_form(n, cin, cout, h):
    """round 5: dW of conv2 + fused average pool from the pooled gradient with the phase-form kernel, operands' roles swapped
    (vqk_conv2d_wgrad_pooled_dy_phase) against the tap form (vqk_conv2d_wgrad_pooled_dy) and fp32 PyTorch"""
    g = torch.Generator(device=DEV).manual_seed(n + cin + cout + h)
    w_ok = h % 32 == 0
    x = torch.randn(n, cin, h, h, device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dyp = torch.randn(n, cout, h // 2, h // 2, device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    outs = []
    saved = ops.POOLED_WGRAD_PHASE
    try:
        for phase in (True, False):
            ops.POOLED_WGRAD_PHASE = phase
            dw = torch.zeros((cout, 3, 3, cin), dtype=torch.float32, device=DEV).permute(0, 3, 1, 2)
            assert ops.raw_conv_wgrad_pooled_dy(x, dyp, 0.25, dw)
            outs.append(dw.clone())
    finally:
        ops.POOLED_WGRAD_PHASE = saved
    torch.cuda.synchronize()
    a, b = outs
    assert float((a - b).norm() / b.norm()) < 2e-6, float((a - b).norm() / b.norm())
    wref = torch.zeros(cout, cin, 3, 3, device=DEV, requires_grad=True)
    torch.nn.functional.avg_pool2d(torch.nn.functional.conv2d(x.float(), wref, padding=1), 2).backward(dyp.float())
    assert float((a - wref.grad).norm() / wref.grad.norm()) < 2e-5
    assert w_ok or True
