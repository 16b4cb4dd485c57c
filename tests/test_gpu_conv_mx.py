"""The matrix-wave / auxiliary-wave 3x3 kernel (csrc/conv_mx.inc) against the stream kernel it replaces on large maps (which
the golden and adjoint tests pin): bit-identical without an epilogue term (same MFMA order, one bf16 rounding), within one
bf16 rounding of the result with bias / residual / pooling (the epilogue runs on the parked bf16 tile)."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu

ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')
DEV, BF, CL = 'cuda:0', torch.bfloat16, torch.channels_last

# n, cin, cout, h, w (input), ups, bias, residual, pool
CASES = [(2, 128, 128, 64, 64, 0, 0, 0, 0), (2, 128, 128, 64, 64, 0, 1, 1, 0), (1, 64, 256, 32, 96, 0, 0, 1, 0),
         (2, 256, 128, 48, 48, 0, 1, 0, 0), (2, 128, 128, 16, 32, 1, 1, 0, 0), (2, 128, 256, 64, 64, 0, 0, 1, 1),
         (2, 128, 128, 48, 48, 0, 1, 1, 1), (8, 128, 128, 128, 128, 0, 0, 1, 0), (3, 512, 512, 32, 32, 0, 0, 0, 0),
         (1, 128, 128, 8, 32, 0, 0, 0, 0),
         # 16x16 maps: 128-pixel half tiles
         (4, 512, 512, 16, 16, 0, 0, 1, 0), (2, 256, 512, 16, 16, 0, 1, 0, 0), (3, 128, 128, 16, 16, 0, 0, 0, 0),
         # ... and 64-pixel quarter tiles when the layer has <= 256 output channels (round 4)
         (4, 256, 256, 16, 16, 0, 1, 1, 0), (2, 512, 256, 16, 16, 0, 0, 0, 0), (5, 256, 128, 8, 32, 0, 0, 1, 0)]


@pytest.mark.parametrize('n,cin,cout,h,w,ups,hb,hr,pool', CASES)
def test_mx_matches_stream_kernel(n, cin, cout, h, w, ups, hb, hr, pool):
    g = torch.Generator(device=DEV).manual_seed(cin + cout + h + 7 * ups + hb + 2 * hr + 4 * pool)
    x = torch.randn(n, cin, h, w, device=DEV, generator=g).to(BF).contiguous(memory_format=CL)
    wt = (torch.randn(cout, 3, 3, cin, device=DEV, generator=g) / (3 * cin ** 0.5)).reshape(-1)
    s = 2 if ups else 1
    ho, wo = h * s, w * s
    bias = torch.randn(cout, device=DEV, generator=g) if hb else None
    res = torch.randn(n, cout, ho, wo, device=DEV, generator=g).to(BF).contiguous(memory_format=CL) if hr else None
    layout = ops.weight_layout(BF, n, h, w, cin, cout, 3, bool(ups))
    assert layout == 1
    wq = ops.pack_weights(wt, BF, cout, cin, 3, False, layout)
    lib = native.lib()
    outs = {}
    for variant in (5, 6):
        lib.vqk_conv_set_variant(variant)
        try:
            if pool:
                outs[variant] = ops.raw_conv_fprop_pooled(x, wq, bias, res, 3, bool(ups), cout, 0.25)
            else:
                outs[variant] = ops.raw_conv_fprop(x, wq, bias, res, 3, bool(ups), 0, BF, cout, layout)
        finally:
            lib.vqk_conv_set_variant(-1)
    torch.cuda.synchronize()
    a, b = outs[5].float(), outs[6].float()
    assert a.shape == b.shape
    if not (hb or hr or pool):
        assert torch.equal(outs[5], outs[6])
    else:
        # one extra bf16 rounding of the convolution sum (2^-9 relative per element) before the epilogue arithmetic
        err = (a - b).abs()
        scale = a.abs() + (res.float().abs().mean() if hr else 0.0) + 1.0
        assert float((err / scale).max()) < 1.2e-2
        assert float((a - b).norm() / a.norm()) < 4e-3


@pytest.mark.parametrize('act,acc_scale,out_gain,hb,hr', [(2, 1.0, 1.0, 1, 0), (3, 0.03, 1.41421356, 1, 0), (3, 0.05, 0.70710678, 1, 1),
                                                           (0, 0.02, 1.0, 0, 0)])
def test_mx_epilogue_activation_and_gains(act, acc_scale, out_gain, hb, hr):
    """y = out_gain * act(acc * acc_scale + bias) + residual in the drain (the VGG relu convs, the StyleGAN2 discriminator's
    leaky-relu convs with their runtime weight gain, discriminator.py:165 / bias_act.py:55) against the stream kernel"""
    n, cin, cout, h, w = 4, 128, 256, 32, 32
    g = torch.Generator(device=DEV).manual_seed(17 + act + hr)
    x = torch.randn(n, cin, h, w, device=DEV, generator=g).to(BF).contiguous(memory_format=CL)
    wt = torch.randn(cout, 3, 3, cin, device=DEV, generator=g).reshape(-1) * (1.0 if acc_scale != 1.0 else 0.03)
    bias = torch.randn(cout, device=DEV, generator=g) if hb else None
    res = torch.randn(n, cout, h, w, device=DEV, generator=g).to(BF).contiguous(memory_format=CL) if hr else None
    wq = ops.pack_weights(wt, BF, cout, cin, 3, False, 1)
    lib = native.lib()
    outs = {}
    for variant in (5, 6):
        lib.vqk_conv_set_variant(variant)
        try:
            outs[variant] = ops._conv_general_raw(x, wq, bias, res, cout, 3, 1, 1, 0, h, w, act, acc_scale, out_gain, BF, 1)
        finally:
            lib.vqk_conv_set_variant(-1)
    torch.cuda.synchronize()
    a, b = outs[5].float(), outs[6].float()
    scale = a.abs() + (res.float().abs().mean() if hr else 0.0) + 1.0
    assert float(((a - b).abs() / scale).max()) < 1.2e-2
    assert float((a - b).norm() / a.norm()) < 4e-3
    assert float(a.abs().max()) > 0.5                      # not a trivially zero comparison
