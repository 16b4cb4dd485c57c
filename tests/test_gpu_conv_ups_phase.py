"""The nearest-x2 upsample + 3x3 conv (autoencoder.py:104-106) in PHASE form -- four 2x2-tap convs on the low-resolution input
with pre-summed weights, vqk_conv2d_ups_phase -- against the same conv evaluated tap by tap over the upsampled image (the
stream kernels, which the golden tests pin to the reference's Upsample): forward, data gradient, fused GroupNorm sums."""
import importlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
DEV, BF, CL = 'cuda:0', torch.bfloat16, torch.channels_last

# n, cin, cout, h, w (low resolution), bias
CASES = [(2, 128, 128, 16, 32, 1), (1, 128, 256, 8, 32, 0), (3, 256, 128, 16, 16, 1), (2, 128, 128, 64, 64, 1), (1, 256, 256, 24, 32, 0)]


def _ref(x, wt, bias, dy=None):
    xu = x.float().repeat_interleave(2, dim=2).repeat_interleave(2, dim=3).requires_grad_(dy is not None)
    y = F.conv2d(xu, wt, bias, padding=1)
    if dy is None:
        return y
    (dxu,) = torch.autograd.grad(y, xu, dy.float())
    return y, F.avg_pool2d(dxu, 2) * 4.0


@pytest.mark.parametrize('n,cin,cout,h,w,hb', CASES)
def test_phase_form_matches_tap_form(n, cin, cout, h, w, hb):
    g = torch.Generator(device=DEV).manual_seed(cin + cout + h + w + hb)
    x = torch.randn(n, cin, h, w, device=DEV, generator=g).to(BF).contiguous(memory_format=CL)
    wt = (torch.randn(cout, cin, 3, 3, device=DEV, generator=g) / (3 * cin ** 0.5)).contiguous(memory_format=CL)
    bias = torch.randn(cout, device=DEV, generator=g) if hb else None
    dy = torch.randn(n, cout, 2 * h, 2 * w, device=DEV, generator=g).to(BF).contiguous(memory_format=CL)
    wmem = wt.permute(0, 2, 3, 1).reshape(-1)                       # [Cout][kh][kw][Cin]
    w4 = ops.pack_weights(wmem, BF, cout, cin, 3, False, 2)
    w4t = ops.pack_weights(wmem, BF, cout, cin, 3, True, 2)
    y = ops.raw_conv_ups_phase(x, w4, bias, cout, False)
    assert y is not None, 'the phase kernel must serve this shape'
    dx = ops.raw_conv_ups_phase(dy, w4t, None, cin, True)
    assert dx is not None
    y_ref, dx_ref = _ref(x, wt.to(BF).float(), bias, dy)           # exact conv on the bf16-rounded 3x3 weights
    torch.cuda.synchronize()
    # the phase weights are sums of up to four bf16-rounded... fp32 taps rounded ONCE to bf16: a 2^-9 relative
    # perturbation of each summed weight against rounding the nine taps individually
    assert float((y.float() - y_ref).norm() / y_ref.norm()) < 6e-3
    assert float((dx.float() - dx_ref).norm() / dx_ref.norm()) < 6e-3
    # against the tap-by-tap kernels on the same operands: same size of deviation
    lay = ops.weight_layout(BF, n, h, w, cin, cout, 3, True)
    y_tap = ops.raw_conv_fprop(x, ops.pack_weights(wmem, BF, cout, cin, 3, False, lay), bias, None, 3, True, 0, BF, cout, lay)
    assert float((y.float() - y_tap.float()).norm() / y_tap.float().norm()) < 8e-3


def test_phase_form_gn_sums():
    n, c, h, w = 2, 128, 32, 32
    g = torch.Generator(device=DEV).manual_seed(5)
    x = (torch.randn(n, c, h, w, device=DEV, generator=g) + 0.3).to(BF).contiguous(memory_format=CL)
    wmem = (torch.randn(c, 3, 3, c, device=DEV, generator=g) / (3 * c ** 0.5)).reshape(-1)
    bias = torch.randn(c, device=DEV, generator=g)
    gw, gb = torch.randn(c, device=DEV, generator=g), torch.randn(c, device=DEV, generator=g)
    w4 = ops.pack_weights(wmem, BF, c, c, 3, False, 2)
    y = ops.raw_conv_ups_phase(x, w4, bias, c, False, gn_groups=32)
    assert y is not None and ops.pending_gn() is not None
    a, st = ops.raw_gn_forward(y, gw, gb, 32, 1e-6, True)           # claims the sums
    a_ref, st_ref = ops.raw_gn_forward(y.clone(), gw, gb, 32, 1e-6, True)
    torch.cuda.synchronize()
    torch.testing.assert_close(st.view(-1, 2)[:, 0], st_ref.view(-1, 2)[:, 0], rtol=0, atol=2e-5)
    torch.testing.assert_close(st.view(-1, 2)[:, 1], st_ref.view(-1, 2)[:, 1], rtol=2e-5, atol=0)
    assert float(ops._gn_ws(x.device, n * 64 + n).abs().max()) == 0.0


@pytest.mark.parametrize('n,cin,cout,h,w', [(2, 128, 128, 16, 32), (4, 256, 256, 32, 32), (1, 64, 128, 24, 48), (3, 128, 64, 8, 16),
                                             (2, 128, 128, 64, 64)])
def test_phase_form_weight_gradient_matches_tap_form_and_torch(n, cin, cout, h, w):
    """round 5: dW of nearest-x2 + 3x3 conv (autoencoder.py:102-105) as four 2x2-window launches on the low-resolution input
    (vqk_conv2d_wgrad_ups_phase) against the tap form (the 3x3 kernel reading x through the upsample addressing) -- same bf16
    operands, fp32 accumulation: only the summation order differs -- and against fp32 PyTorch on the same operands"""
    g = torch.Generator(device=DEV).manual_seed(n + cin + cout + h)
    x = torch.randn(n, cin, h, w, device=DEV, generator=g).to(BF).contiguous(memory_format=CL)
    dy = torch.randn(n, cout, 2 * h, 2 * w, device=DEV, generator=g).to(BF).contiguous(memory_format=CL)
    saved = ops.UPS_PHASE_WGRAD
    try:
        ops.UPS_PHASE_WGRAD = True
        a = ops.raw_conv_wgrad(x, dy, 3, True).float().clone()
        ops.UPS_PHASE_WGRAD = False
        b = ops.raw_conv_wgrad(x, dy, 3, True).float().clone()
    finally:
        ops.UPS_PHASE_WGRAD = saved
    torch.cuda.synchronize()
    assert a.shape == b.shape == (cout, cin, 3, 3)
    assert float((a - b).norm() / b.norm()) < 2e-6
    xu = torch.nn.functional.interpolate(x.float(), scale_factor=2, mode='nearest')
    wref = torch.zeros(cout, cin, 3, 3, device=DEV, requires_grad=True)
    torch.nn.functional.conv2d(xu, wref, padding=1).backward(dy.float())
    assert float((a - wref.grad).norm() / wref.grad.norm()) < 2e-5
    # accumulation into an existing buffer (the optimizer arena): dw += gradient
    base = torch.randn(cout, 3, 3, cin, device=DEV, generator=g).permute(0, 3, 1, 2)
    acc = base.clone()
    ops.raw_conv_wgrad(x, dy, 3, True, out=acc)
    assert float((acc - base - a).norm() / a.norm()) < 2e-6
