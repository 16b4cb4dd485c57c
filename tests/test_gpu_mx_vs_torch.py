"""The role-split bf16 conv kernels (conv3x3_mx_kernel: fprop / dgrad / phase-form upsample; conv3x3_wgrad_mx_kernel: weight
gradient, plain and with a pooled dy) pinned DIRECTLY to plain PyTorch fp32 ``F.conv2d`` on the CPU -- not to the stream
kernel they replaced -- at shapes only these kernels serve (>= 64x64 maps, 128 / 256 channels).  Operands are bf16-exact, so
the only differences are the fp32 accumulation order and the bf16 roundings of the result the kernel documents
(csrc/conv_mx.hip header: one rounding of the sum, one more after bias / residual / pooling).
Reference semantics: vqvae/modules/autoencoder.py:63-77 (ResBlock convs), :89-91 (avg-pool), :102-105 (Upsample)."""
import importlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
DEV, BF, CL = 'cuda:0', torch.bfloat16, torch.channels_last
BF_EPS = 2.0 ** -8          # one bf16 rounding: <= 2^-9 relative; two roundings <= 2^-8


def _bf(t):
    return t.to(BF).float()


def _dev(t):
    return t.to(DEV).to(BF).contiguous(memory_format=CL)


def _check(got, want, roundings=1, what=''):
    got, want = got.float().cpu(), want.float()
    assert got.shape == want.shape, (got.shape, want.shape)
    rel = float((got - want).norm() / want.norm())
    # elementwise: bf16 rounding is relative to the element; sums of terms of mixed sign also carry the rounding of the
    # intermediate (the parked conv sum), bounded by the typical magnitude
    tol = roundings * BF_EPS * (want.abs() + want.abs().mean())
    worst = float(((got - want).abs() / tol).max())
    assert rel < 3e-3 * roundings, (what, rel)
    assert worst < 1.0, (what, worst)


def _events():
    ops.KERNEL_EVENTS = []


def _kernels():
    ev, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    return [e[0] for e in ev]


@pytest.mark.parametrize('n,cin,cout,h,w', [(2, 128, 128, 128, 128), (1, 256, 256, 64, 96)])
def test_mx_fprop_bias_residual_pool_vs_torch(n, cin, cout, h, w):
    g = torch.Generator().manual_seed(cin + h)
    x = _bf(torch.randn(n, cin, h, w, generator=g))
    wt = _bf(torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5))
    bias = torch.randn(cout, generator=g)
    res = _bf(torch.randn(n, cout, h, w, generator=g))
    want_full = F.conv2d(x, wt, bias, padding=1) + res
    want_pool = F.avg_pool2d(want_full, 2)
    wmem = wt.permute(0, 2, 3, 1).contiguous().reshape(-1).to(DEV)
    assert ops.weight_layout(BF, n, h, w, cin, cout, 3, False) == 1
    wq = ops.pack_weights(wmem, BF, cout, cin, 3, False, 1)
    _events()
    y_full = ops.raw_conv_fprop(_dev(x), wq, bias.to(DEV), _dev(res), 3, False, 0, BF, cout, 1)
    y_pool = ops.raw_conv_fprop_pooled(_dev(x), wq, bias.to(DEV), _dev(res), 3, False, cout, 0.25)
    torch.cuda.synchronize()
    assert _kernels() == ['conv3x3_mx_kernel<bf16>'] * 2
    _check(y_full, want_full, 2, 'bias+residual')
    _check(y_pool, want_pool, 2, 'bias+residual+pool')


def test_mx_dgrad_and_wgrad_vs_torch_autograd():
    """conv2d autograd node at a ResBlock shape: dx (the mx kernel on the transposed operand) and dW (conv3x3_wgrad_mx_kernel)
    against torch autograd of F.conv2d in fp32"""
    n, cin, cout, h, w = 2, 128, 256, 64, 64
    g = torch.Generator().manual_seed(5)
    x = _bf(torch.randn(n, cin, h, w, generator=g)).requires_grad_(True)
    wt = _bf(torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).requires_grad_(True)
    dy = _bf(torch.randn(n, cout, h, w, generator=g))
    F.conv2d(x, wt, None, padding=1).backward(dy)
    xd = _dev(x.detach()).requires_grad_(True)
    wd = wt.detach().to(DEV).contiguous(memory_format=CL).requires_grad_(True)
    _events()
    y = ops.conv2d(xd, wd)
    y.backward(_dev(dy))
    torch.cuda.synchronize()
    names = _kernels()
    assert names.count('conv3x3_mx_kernel<bf16>') == 2 and names.count('conv3x3_wgrad_mx_kernel<bf16>') == 1, names
    _check(y.detach(), F.conv2d(x.detach(), wt.detach(), None, padding=1), 1, 'fprop')
    _check(xd.grad, x.grad, 1, 'dgrad')
    gw, want = wd.grad.float().cpu(), wt.grad
    assert float((gw - want).norm() / want.norm()) < 2e-5           # fp32 accumulation of exact bf16 products
    assert float((gw - want).abs().max() / want.abs().max()) < 1e-4


def test_phase_form_upsample_fwd_and_dgrad_vs_torch():
    """nearest x2 + 3x3 as four 2x2-tap launches with pre-summed weights (forward and data gradient) against
    F.conv2d(F.interpolate(x, 2, 'nearest')) and its autograd; the pre-summed weights are rounded to bf16 once, so the
    tolerance carries one extra rounding of the operand"""
    n, c, h, w = 2, 128, 64, 64
    g = torch.Generator().manual_seed(9)
    x = _bf(torch.randn(n, c, h, w, generator=g)).requires_grad_(True)
    wt = _bf(torch.randn(c, c, 3, 3, generator=g) / (3 * c ** 0.5)).requires_grad_(True)
    bias = torch.randn(c, generator=g).requires_grad_(True)
    dy = _bf(torch.randn(n, c, 2 * h, 2 * w, generator=g))
    want = F.conv2d(F.interpolate(x, scale_factor=2.0, mode='nearest-exact'), wt, bias, padding=1)
    want.backward(dy)
    xd = _dev(x.detach()).requires_grad_(True)
    wd = wt.detach().to(DEV).contiguous(memory_format=CL).requires_grad_(True)
    bd = bias.detach().to(DEV).requires_grad_(True)
    _events()
    y = ops.conv2d(xd, wd, bd, ups=True)
    y.backward(_dev(dy))
    torch.cuda.synchronize()
    names = _kernels()
    assert names.count('conv3x3_mx_kernel<bf16>') == 2, names     # forward + data gradient, both in phase form
    got, ref = y.detach().float().cpu(), want.detach()
    assert float((got - ref).norm() / ref.norm()) < 4e-3
    gx = xd.grad.float().cpu()
    assert float((gx - x.grad).norm() / x.grad.norm()) < 6e-3       # four phases accumulate through bf16
    gw = wd.grad.float().cpu()
    assert float((gw - wt.grad).norm() / wt.grad.norm()) < 2e-5
    assert float((bd.grad.cpu() - bias.grad).norm() / bias.grad.norm()) < 1e-5


def test_wgrad_pooled_dy_vs_torch():
    """dW += scale * wgrad(x, unpool(dy_pooled)) without the unpooled tensor, against torch autograd through avg_pool2d"""
    n, c, h, w = 2, 128, 64, 64
    g = torch.Generator().manual_seed(11)
    x = _bf(torch.randn(n, c, h, w, generator=g))
    wt = torch.zeros(c, c, 3, 3, requires_grad=True)
    dyp = _bf(torch.randn(n, c, h // 2, w // 2, generator=g))
    F.avg_pool2d(F.conv2d(x, wt, None, padding=1), 2).backward(dyp)
    out = torch.zeros((c, 3, 3, c), dtype=torch.float32, device=DEV).permute(0, 3, 1, 2)
    _events()
    assert ops.raw_conv_wgrad_pooled_dy(_dev(x), _dev(dyp), 0.25, out)
    torch.cuda.synchronize()
    assert _kernels() == ['conv3x3_wgrad_mx_kernel<bf16>']
    got = out.float().cpu()
    assert float((got - wt.grad).norm() / wt.grad.norm()) < 2e-5


@pytest.mark.parametrize('cin,cout', [(8, 128), (128, 8)])
@pytest.mark.parametrize('n,h,w', [(2, 64, 64), (3, 24, 32), (1, 256, 256), (2, 16, 128)])
def test_edge_wgrad_thin_kernel_vs_torch(cin, cout, n, h, w):
    """the K = 72 weight-gradient kernel of the two edge convs (conv_edge.hip; autoencoder.py:114 conv_in, :170 conv_out with
    the 3 image channels padded to 8) against torch autograd of F.conv2d in fp32; its workspace split-K is deterministic"""
    g = torch.Generator().manual_seed(cin + h)
    x = _bf(torch.randn(n, cin, h, w, generator=g))
    dy = _bf(torch.randn(n, cout, h, w, generator=g))
    wt = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    F.conv2d(x, wt, None, padding=1).backward(dy)
    outs = []
    for _ in range(2):
        _events()
        dw = ops.raw_conv_wgrad(_dev(x), _dev(dy), 3, False)
        torch.cuda.synchronize()
        assert _kernels() == ['conv3x3_wgrad_thin_kernel<bf16>']
        outs.append(dw.float().cpu())
    assert torch.equal(outs[0], outs[1])                             # ordered reduce: bit-identical run to run
    want = wt.grad
    assert outs[0].shape == want.shape
    assert float((outs[0] - want).norm() / want.norm()) < 2e-5
    assert float((outs[0] - want).abs().max() / want.abs().max()) < 1e-4
    # accumulates into a pre-existing buffer
    base = torch.ones((cout, 3, 3, cin), dtype=torch.float32, device=DEV).permute(0, 3, 1, 2)
    ops.raw_conv_wgrad(_dev(x), _dev(dy), 3, False, out=base)
    torch.cuda.synchronize()
    assert float((base.float().cpu() - 1.0 - want).norm() / want.norm()) < 2e-5


@pytest.mark.parametrize('n,cin,cout,h,w', [(2, 128, 256, 64, 64), (3, 256, 128, 32, 64), (2, 256, 512, 16, 16), (1, 512, 256, 16, 32)])
def test_conv1x1_on_mx_kernel_vs_torch(n, cin, cout, h, w):
    """the ResBlock shortcut convs (autoencoder.py:52-55: 1x1, no bias) on the NTAP = 1 form of the matrix/auxiliary-wave
    kernel: forward and data gradient against F.conv2d / its autograd in fp32"""
    g = torch.Generator().manual_seed(cin + cout + h)
    x = _bf(torch.randn(n, cin, h, w, generator=g)).requires_grad_(True)
    wt = _bf(torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).requires_grad_(True)
    dy = _bf(torch.randn(n, cout, h, w, generator=g))
    want = F.conv2d(x, wt)
    want.backward(dy)
    xd = _dev(x.detach()).requires_grad_(True)
    wd = wt.detach().to(DEV).contiguous(memory_format=CL).requires_grad_(True)
    _events()
    y = ops.conv2d(xd, wd)
    y.backward(_dev(dy))
    torch.cuda.synchronize()
    names = _kernels()
    assert names.count('conv1x1_mx_kernel<bf16> (HBM)') == 2, names  # forward + data gradient
    _check(y.detach(), want.detach(), 1, '1x1 fprop')
    _check(xd.grad, x.grad, 1, '1x1 dgrad')
    gw = wd.grad.float().cpu()
    assert float((gw - wt.grad).norm() / wt.grad.norm()) < 2e-5


@pytest.mark.parametrize('n,h,w,act', [(2, 64, 64, 1), (1, 256, 256, 1), (3, 8, 32, 0), (2, 24, 96, 1)])
def test_thin_out_head_conv_vs_torch(n, h, w, act):
    """the decoder's last conv (autoencoder.py:170: 128 -> 3 channels padded to 8, + bias + tanh) on the 16x16x32-MFMA kernel of
    csrc/conv_edge.hip against F.conv2d in fp32, through the product's Conv2dFn (forward; its backward kernels are pinned above)"""
    g = torch.Generator().manual_seed(h + w)
    x = _bf(torch.randn(n, 128, h, w, generator=g))
    wt = _bf(torch.randn(3, 128, 3, 3, generator=g) / 34.0)
    bias = torch.randn(3, generator=g) * 0.1
    want = F.conv2d(x, wt, bias, padding=1)
    if act == 1:
        want = torch.tanh(want)
    wd = wt.to(DEV).contiguous(memory_format=CL)
    _events()
    y = ops.conv2d(_dev(x), wd, bias.to(DEV), act=act)
    torch.cuda.synchronize()
    assert _kernels() == ['conv3x3_thin_out_kernel<bf16> (HBM)']
    assert y.shape[1] == 8 and float(y[:, 3:].float().abs().max()) == 0.0          # pad channels stay zero
    _check(y[:, :3], want, 1, 'thin-out head')
