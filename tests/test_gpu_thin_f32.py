"""The exact-fp32 kernels of the two EDGE convs (csrc/conv_thin_f32.hip: 3 -> C on the padded 4-channel image, C -> 3 with bias +
tanh; autoencoder.py:132 / :170) in the fp32 compute modes, against fp64 torch convolutions and against the general implicit-GEMM
kernel they replace (vqk_conv_set_variant(0))."""
import importlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')
DEV, F32, CL = 'cuda:0', torch.float32, torch.channels_last
TOL = 2e-6


def _rel(a, ref):
    return float((a.double() - ref).abs().max()) / float(ref.abs().max())


@pytest.mark.parametrize('n,c,h,w,act,hb,hr', [(2, 128, 16, 32, 1, 1, 0), (1, 64, 8, 64, 0, 0, 1), (3, 32, 24, 32, 0, 1, 1), (2, 128, 64, 64, 1, 1, 0)])
def test_thin_out_forward(n, c, h, w, act, hb, hr):
    g = torch.Generator(device=DEV).manual_seed(c + h + act)
    x = torch.randn(n, c, h, w, device=DEV, generator=g).contiguous(memory_format=CL)
    wt = torch.zeros(4, 3, 3, c, device=DEV)
    wt[:3] = torch.randn(3, 3, 3, c, device=DEV, generator=g) / (3 * c ** 0.5)
    bias = torch.zeros(4, device=DEV)
    bias[:3] = torch.randn(3, device=DEV, generator=g)
    res = torch.randn(n, 4, h, w, device=DEV, generator=g).contiguous(memory_format=CL) if hr else None
    assert ops.weight_layout(F32, n, h, w, c, 4, 3, False) == 0
    y = ops.raw_conv_fprop(x, wt.reshape(-1), bias if hb else None, res, 3, False, act, F32, 4, 0)
    ref = F.conv2d(x.double(), wt.permute(0, 3, 1, 2).double(), bias.double() if hb else None, padding=1)
    ref = torch.tanh(ref) if act else ref
    if hr:
        ref = ref + res.double()
    native.lib().vqk_conv_set_variant(0)
    try:
        y0 = ops.raw_conv_fprop(x, wt.reshape(-1), bias if hb else None, res, 3, False, act, F32, 4, 0)
    finally:
        native.lib().vqk_conv_set_variant(-1)
    torch.cuda.synchronize()
    tol = 1e-5 if act else TOL                             # (the device tanhf is a few 1e-6 off near saturation, in both kernels)
    assert _rel(y, ref) < tol and _rel(y0, ref) < tol


@pytest.mark.parametrize('n,c,h,w,hb', [(2, 128, 16, 32, 0), (1, 64, 9, 20, 1), (2, 256, 8, 16, 1), (2, 128, 64, 64, 0)])
def test_thin_in_forward_and_as_data_gradient(n, c, h, w, hb):
    g = torch.Generator(device=DEV).manual_seed(c + h)
    x = torch.zeros(n, 4, h, w, device=DEV).contiguous(memory_format=CL)
    x[:, :3] = torch.randn(n, 3, h, w, device=DEV, generator=g)
    wt = torch.zeros(c, 3, 3, 4, device=DEV)
    wt[..., :3] = torch.randn(c, 3, 3, 3, device=DEV, generator=g) / 5
    bias = torch.randn(c, device=DEV, generator=g) if hb else None
    y = ops.raw_conv_fprop(x, wt.reshape(-1), bias, None, 3, False, 0, F32, c, 0)
    ref = F.conv2d(x.double(), wt.permute(0, 3, 1, 2).double(), bias.double() if hb else None, padding=1)
    torch.cuda.synchronize()
    assert _rel(y, ref) < TOL
    # the data gradient of the C -> 4 conv: dy [n, 4, h, w] with the transposed / flipped operand
    w_out = torch.zeros(4, 3, 3, c, device=DEV)
    w_out[:3] = torch.randn(3, 3, 3, c, device=DEV, generator=g) / (3 * c ** 0.5)
    dy = torch.randn(n, 4, h, w, device=DEV, generator=g).contiguous(memory_format=CL)
    wtr = ops.pack_weights(w_out.reshape(-1), F32, 4, c, 3, True, 0)
    dx = ops.raw_conv_fprop(dy, wtr, None, None, 3, False, 0, F32, c, 0)
    refd = F.conv_transpose2d(dy.double(), w_out.permute(0, 3, 1, 2).double(), padding=1)
    torch.cuda.synchronize()
    assert _rel(dx, refd) < TOL


@pytest.mark.parametrize('n,c,h,w', [(2, 128, 16, 32), (3, 64, 9, 20), (2, 256, 8, 16), (4, 128, 64, 64)])
def test_thin_weight_gradients(n, c, h, w):
    g = torch.Generator(device=DEV).manual_seed(2 * c + h)
    thin = torch.zeros(n, 4, h, w, device=DEV).contiguous(memory_format=CL)
    thin[:, :3] = torch.randn(n, 3, h, w, device=DEV, generator=g)
    wide = torch.randn(n, c, h, w, device=DEV, generator=g).contiguous(memory_format=CL)
    # conv_in: x thin, dy wide -> dW [c][3][3][4]
    for x, dy, co, ci in ((thin, wide, c, 4), (wide, thin, 4, c)):
        pre = torch.randn(co, 3, 3, ci, device=DEV, generator=g).permute(0, 3, 1, 2)
        dw = ops.raw_conv_wgrad(x, dy, 3, False, out=pre.clone(memory_format=torch.preserve_format))
        wd = torch.zeros(co, ci, 3, 3, device=DEV, dtype=torch.float64, requires_grad=True)
        (F.conv2d(x.double(), wd, None, padding=1) * dy.double()).sum().backward()
        ref = wd.grad + pre.double()
        native.lib().vqk_conv_set_variant(0)
        try:
            dw0 = ops.raw_conv_wgrad(x, dy, 3, False, out=pre.clone(memory_format=torch.preserve_format))
        finally:
            native.lib().vqk_conv_set_variant(-1)
        torch.cuda.synchronize()
        assert _rel(dw, ref) < 1e-5, (co, ci)
        assert _rel(dw0, ref) < 1e-5
