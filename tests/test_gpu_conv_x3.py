"""The split-product ("bf16x3") 3x3 convolution (csrc/conv_x3.hip, include/vqk.h layout 5): fp32 activations in / out, every
product as x_hi w_hi + x_lo w_hi + x_hi w_lo on the bf16 matrix pipe.  Checked against the exact-fp32 kernel of the parity mode
(v_mfma_f32_32x32x2_f32, itself pinned by the reference fixtures in tests/test_gpu_ops.py) and against an fp64 torch convolution:
the error has to sit at the 2^-17-per-product level -- two orders below the plain bf16 mode, one above fp32 rounding."""
import importlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')
DEV, F32, BF, CL = 'cuda:0', torch.float32, torch.bfloat16, torch.channels_last
TOL = 3e-5          # relative to the largest output magnitude; the plain bf16 kernels sit at ~3e-3 on the same problems


def _ref64(x, w_okki, bias, res, ups):
    xd = x.double()
    if ups:
        xd = F.interpolate(xd, scale_factor=2, mode='nearest')
    y = F.conv2d(xd, w_okki.permute(0, 3, 1, 2).double(), bias.double() if bias is not None else None, padding=1)
    return y + res.double() if res is not None else y


# n, cin, cout, h, w (input), ups, bias, residual
CASES = [(2, 128, 128, 32, 32, 0, 0, 0), (2, 128, 128, 32, 32, 0, 1, 1), (1, 64, 256, 16, 48, 0, 0, 1), (2, 256, 128, 24, 32, 0, 1, 0),
         (2, 128, 128, 16, 16, 1, 1, 0), (3, 512, 512, 16, 16, 0, 0, 1), (1, 96, 64, 8, 16, 0, 1, 1), (2, 32, 32, 8, 32, 0, 0, 0),
         (5, 128, 256, 40, 16, 0, 0, 0), (2, 256, 512, 8, 16, 1, 0, 0)]


@pytest.mark.parametrize('wl', [0, 1])
@pytest.mark.parametrize('n,cin,cout,h,w,ups,hb,hr', CASES)
def test_x3_fprop_matches_exact_fp32(n, cin, cout, h, w, ups, hb, hr, wl):
    g = torch.Generator(device=DEV).manual_seed(cin + 3 * cout + h + 7 * ups + hb + 2 * hr)
    x = torch.randn(n, cin, h, w, device=DEV, generator=g).contiguous(memory_format=CL)
    wt = torch.randn(cout, 3, 3, cin, device=DEV, generator=g) / (3 * cin ** 0.5)
    s = 2 if ups else 1
    bias = torch.randn(cout, device=DEV, generator=g) if hb else None
    res = torch.randn(n, cout, h * s, w * s, device=DEV, generator=g).contiguous(memory_format=CL) if hr else None
    assert ops.weight_layout(F32, n, h, w, cin, cout, 3, bool(ups), x3=True) == 5
    assert ops.weight_layout(F32, n, h, w, cin, cout, 3, bool(ups), x3=False) in (0, 1)
    w5 = ops.pack_weights(wt.reshape(-1), F32, cout, cin, 3, False, 5)
    native.lib().vqk_set_tuning(b'X3_WL', wl)
    try:
        y3 = ops.raw_conv_fprop(x, w5, bias, res, 3, bool(ups), 0, F32, cout, 5)
    finally:
        native.lib().vqk_set_tuning(b'X3_WL', 1)
    l1 = ops.weight_layout(F32, n, h, w, cin, cout, 3, bool(ups), x3=False)
    y1 = ops.raw_conv_fprop(x, ops.pack_weights(wt.reshape(-1), F32, cout, cin, 3, False, l1), bias, res, 3, bool(ups), 0, F32, cout, l1)
    ref = _ref64(x, wt, bias, res, ups)
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    e3 = float((y3.double() - ref).abs().max()) / scale
    e1 = float((y1.double() - ref).abs().max()) / scale
    assert e1 < 5e-6, e1                                   # the exact-fp32 kernel: fp32 summation noise only
    assert e3 < TOL, (e3, e1)
    assert float((y3.double() - ref).norm() / ref.norm()) < 1e-5


def test_x3_is_two_orders_better_than_bf16():
    n, c, h, w = 2, 128, 32, 32
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(n, c, h, w, device=DEV, generator=g).contiguous(memory_format=CL)
    wt = torch.randn(c, 3, 3, c, device=DEV, generator=g) / (3 * c ** 0.5)
    ref = _ref64(x, wt, None, None, 0)
    y3 = ops.raw_conv_fprop(x, ops.pack_weights(wt.reshape(-1), F32, c, c, 3, False, 5), None, None, 3, False, 0, F32, c, 5)
    lb = ops.weight_layout(BF, n, h, w, c, c, 3, False)
    yb = ops.raw_conv_fprop(x.to(BF).contiguous(memory_format=CL), ops.pack_weights(wt.reshape(-1), BF, c, c, 3, False, lb), None, None,
                            3, False, 0, BF, c, lb)
    torch.cuda.synchronize()
    e3 = float((y3.double() - ref).norm() / ref.norm())
    eb = float((yb.double() - ref).norm() / ref.norm())
    assert e3 * 100 < eb, (e3, eb)


@pytest.mark.parametrize('act,acc_scale,out_gain,hb,hr,cout', [(2, 1.0, 1.0, 1, 0, 128), (3, 0.03, 1.41421356, 1, 1, 128), (0, 1.0, 1.0, 1, 1, 64),
                                                                (1, 1.0, 1.0, 1, 0, 32)])
def test_x3_general_epilogue(act, acc_scale, out_gain, hb, hr, cout):
    """y = out_gain * act(acc * acc_scale + bias) + residual, and cout tiles that are not whole (64, 32 couts)"""
    n, cin, h, w = 2, 64, 16, 32
    g = torch.Generator(device=DEV).manual_seed(11 + act + cout)
    x = torch.randn(n, cin, h, w, device=DEV, generator=g).contiguous(memory_format=CL)
    wt = torch.randn(cout, 3, 3, cin, device=DEV, generator=g) * (1.0 if acc_scale != 1.0 else 0.05)
    bias = torch.randn(cout, device=DEV, generator=g) if hb else None
    res = torch.randn(n, cout, h, w, device=DEV, generator=g).contiguous(memory_format=CL) if hr else None
    y3 = ops._conv_general_raw(x, ops.pack_weights(wt.reshape(-1), F32, cout, cin, 3, False, 5), bias, res, cout, 3, 1, 1, 0, h, w, act,
                               acc_scale, out_gain, F32, 5)
    acc = F.conv2d(x.double(), wt.permute(0, 3, 1, 2).double(), None, padding=1) * acc_scale
    if bias is not None:
        acc = acc + bias.double()[None, :, None, None]
    ref = {0: lambda t: t, 1: torch.tanh, 2: torch.relu, 3: lambda t: F.leaky_relu(t, 0.2)}[act](acc) * out_gain
    if res is not None:
        ref = ref + res.double()
    torch.cuda.synchronize()
    assert float((y3.double() - ref).abs().max()) / float(ref.abs().max()) < TOL


@pytest.mark.parametrize('n,cin,cout,h,w', [(2, 128, 256, 16, 32), (1, 256, 64, 8, 16)])
def test_x3_dgrad_operand(n, cin, cout, h, w):
    """transpose = 1 of layout 5: the data gradient = a conv of dy with the flipped, channel-swapped weights"""
    g = torch.Generator(device=DEV).manual_seed(cin + cout)
    dy = torch.randn(n, cout, h, w, device=DEV, generator=g).contiguous(memory_format=CL)
    wt = torch.randn(cout, 3, 3, cin, device=DEV, generator=g) / (3 * cout ** 0.5)
    wtr = ops.pack_weights(wt.reshape(-1), F32, cout, cin, 3, True, 5)
    dx = ops.raw_conv_fprop(dy, wtr, None, None, 3, False, 0, F32, cin, 5)
    ref = F.conv_transpose2d(dy.double(), wt.permute(0, 3, 1, 2).double(), padding=1)
    torch.cuda.synchronize()
    assert float((dx.double() - ref).abs().max()) / float(ref.abs().max()) < TOL


def test_split_pair_is_hi_lo():
    g = torch.Generator(device=DEV).manual_seed(3)
    t = (torch.randn(2, 64, 8, 16, device=DEV, generator=g) * torch.logspace(-6, 6, 64, device=DEV)[None, :, None, None]).contiguous(memory_format=CL)
    p = ops.raw_split_pair(t)
    torch.cuda.synchronize()
    assert p.shape == (2, 128, 8, 16) and p.dtype == BF
    hi, lo = p[:, :64], p[:, 64:]
    assert torch.equal(hi, t.to(BF))                                           # round-to-nearest-even, like torch
    assert torch.equal(lo, (t - hi.float()).to(BF))
    rel = ((hi.double() + lo.double() - t.double()).abs() / t.double().abs()).max()
    assert float(rel) < 2.0 ** -16


@pytest.mark.parametrize('fold', [False, True])
@pytest.mark.parametrize('n,cin,cout,h,w,ups', [(2, 128, 128, 32, 32, 0), (3, 64, 256, 16, 48, 0), (2, 256, 64, 8, 16, 1), (4, 512, 512, 16, 16, 0),
                                                (5, 128, 64, 24, 16, 0), (1, 64, 64, 8, 8, 1)])
def test_x3_wgrad_matches_fp64(n, cin, cout, h, w, ups, fold, monkeypatch):
    """fold = False: conv3x3_wgrad_x3_kernel (both operands split in registers, three products per staged fragment pair);
    True: the pair-tensor form (vqk_split_pair_f32 twice + the folded bf16 role-split kernel)"""
    monkeypatch.setattr(ops, 'X3_WGRAD_FOLD', fold)
    g = torch.Generator(device=DEV).manual_seed(cin + 2 * cout + h + ups)
    s = 2 if ups else 1
    x = torch.randn(n, cin, h, w, device=DEV, generator=g).contiguous(memory_format=CL)
    dy = torch.randn(n, cout, h * s, w * s, device=DEV, generator=g).contiguous(memory_format=CL)
    pre = torch.randn(cout, 3, 3, cin, device=DEV, generator=g).permute(0, 3, 1, 2)    # dW ACCUMULATES into its target
    dw = pre.clone(memory_format=torch.preserve_format)
    assert dw.permute(0, 2, 3, 1).is_contiguous()
    out = ops.raw_conv_wgrad(x, dy, 3, bool(ups), out=dw, x3=True)
    assert out is dw
    xd = x.double()
    if ups:
        xd = F.interpolate(xd, scale_factor=2, mode='nearest')
    xd.requires_grad_(False)
    wd = torch.zeros(cout, cin, 3, 3, device=DEV, dtype=torch.float64, requires_grad=True)
    (F.conv2d(xd, wd, None, padding=1) * dy.double()).sum().backward()
    ref = wd.grad + pre.double()
    exact = ops.raw_conv_wgrad(x, dy, 3, bool(ups), out=pre.clone(memory_format=torch.preserve_format), x3=False)
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    assert float((exact.double() - ref).abs().max()) / scale < 1e-5
    assert float((dw.double() - ref).abs().max()) / scale < TOL
    assert float((dw.double() - ref).norm() / ref.norm()) < 1e-5


def test_x3_mode_runs_a_resblock_like_the_exact_mode():
    """ops.set_conv_products: the autograd nodes carry the mode of their forward into their backward"""
    n, c, h, w = 2, 128, 16, 32
    g = torch.Generator(device=DEV).manual_seed(9)
    x0 = torch.randn(n, c, h, w, device=DEV, generator=g).contiguous(memory_format=CL)
    ps = [torch.randn(1, c, 1, 1, device=DEV, generator=g) * 0.1 + 1.0, torch.randn(1, c, 1, 1, device=DEV, generator=g) * 0.1,
          (torch.randn(c, c, 3, 3, device=DEV, generator=g) / (3 * c ** 0.5)).contiguous(memory_format=CL),
          torch.randn(1, c, 1, 1, device=DEV, generator=g) * 0.1 + 1.0, torch.randn(1, c, 1, 1, device=DEV, generator=g) * 0.1,
          (torch.randn(c, c, 3, 3, device=DEV, generator=g) / (3 * c ** 0.5)).contiguous(memory_format=CL)]
    dout = torch.randn(n, c, h, w, device=DEV, generator=g).contiguous(memory_format=CL)
    res = {}
    for mode in ('fp32', 'bf16x3'):
        ops.set_conv_products(mode)
        try:
            x = x0.clone(memory_format=torch.preserve_format).requires_grad_(True)
            p = [t.clone(memory_format=torch.preserve_format).requires_grad_(True) for t in ps]
            y = ops.res_block(x, *p)
            ops.set_conv_products('fp32')                 # the backward must not depend on the setting at backward time
            y.backward(dout)
            res[mode] = [y.detach(), x.grad] + [t.grad for t in p]
        finally:
            ops.set_conv_products('fp32')
    torch.cuda.synchronize()
    for a, b in zip(res['fp32'], res['bf16x3']):
        assert float((a - b).abs().max()) / float(a.abs().max()) < 1e-4
    assert not torch.equal(res['fp32'][0], res['bf16x3'][0])      # (it did take the other kernel)


@pytest.mark.parametrize('n,cin,cout,h,w,hb,hr', [(2, 128, 256, 32, 32, 0, 0), (3, 512, 256, 16, 16, 1, 0), (1, 256, 128, 8, 48, 0, 1), (2, 64, 64, 24, 16, 1, 1)])
def test_x3_1x1_forward_and_data_gradient(n, cin, cout, h, w, hb, hr):
    """the NTAP = 1 form (ResBlock shortcuts, autoencoder.py:52-55; the latent's conv_out): forward, and the data gradient through the
    transposed operand"""
    g = torch.Generator(device=DEV).manual_seed(cin + cout + h)
    x = torch.randn(n, cin, h, w, device=DEV, generator=g).contiguous(memory_format=CL)
    wt = torch.randn(cout, 1, 1, cin, device=DEV, generator=g) / cin ** 0.5
    bias = torch.randn(cout, device=DEV, generator=g) if hb else None
    res = torch.randn(n, cout, h, w, device=DEV, generator=g).contiguous(memory_format=CL) if hr else None
    assert ops.weight_layout(F32, n, h, w, cin, cout, 1, False, x3=True) == 5
    assert ops.weight_layout(F32, n, h, w, cin, cout, 1, False, x3=False) == 0
    y = ops.raw_conv_fprop(x, ops.pack_weights(wt.reshape(-1), F32, cout, cin, 1, False, 5), bias, res, 1, False, 0, F32, cout, 5)
    ref = F.conv2d(x.double(), wt.permute(0, 3, 1, 2).double(), bias.double() if hb else None)
    if hr:
        ref = ref + res.double()
    dy = torch.randn(n, cout, h, w, device=DEV, generator=g).contiguous(memory_format=CL)
    dx = ops.raw_conv_fprop(dy, ops.pack_weights(wt.reshape(-1), F32, cout, cin, 1, True, 5), None, None, 1, False, 0, F32, cin, 5)
    refd = F.conv_transpose2d(dy.double(), wt.permute(0, 3, 1, 2).double())
    torch.cuda.synchronize()
    assert float((y.double() - ref).abs().max()) / float(ref.abs().max()) < TOL
    assert float((dx.double() - refd).abs().max()) / float(refd.abs().max()) < TOL


@pytest.mark.parametrize('n,cin,cout,h,w,ups,hb,hr,groups', [(2, 128, 128, 64, 64, 0, 0, 1, 32), (3, 128, 256, 32, 64, 0, 1, 0, 32), (2, 256, 512, 16, 32, 1, 1, 0, 32),
                                                           (1, 64, 128, 40, 48, 0, 0, 0, 8)])
def test_x3_conv_leaves_groupnorm_sums(n, cin, cout, h, w, ups, hb, hr, groups):
    """vqk_conv2d_fprop_x3_gnstats: the conv's epilogue leaves sum / sum of squares per (sample, group) of its fp32 output in the
    stream's GroupNorm workspace (channels per group 4, 8, 16), and the GroupNorm that consumes them equals the unfused sequence"""
    g = torch.Generator(device=DEV).manual_seed(cin + cout + h + groups)
    x = torch.randn(n, cin, h, w, device=DEV, generator=g).contiguous(memory_format=CL)
    wt = torch.randn(cout, 3, 3, cin, device=DEV, generator=g) / (3 * cin ** 0.5)
    s = 2 if ups else 1
    bias = torch.randn(cout, device=DEV, generator=g) if hb else None
    res = torch.randn(n, cout, h * s, w * s, device=DEV, generator=g).contiguous(memory_format=CL) if hr else None
    w5 = ops.pack_weights(wt.reshape(-1), F32, cout, cin, 3, False, 5)
    gw = torch.randn(cout, device=DEV, generator=g) * 0.1 + 1.0
    gb = torch.randn(cout, device=DEV, generator=g) * 0.1
    y = ops.raw_conv_fprop_gnstats(x, w5, bias, res, bool(ups), cout, groups, wlayout=5)
    assert y is not None
    ws = ops._gn_ws(x.device, n * groups * 2 + n)
    torch.cuda.synchronize()
    sums = ws[:n * groups * 2].view(n, groups, 2).clone()
    yd = y.double().view(n, groups, cout // groups, -1)
    ref = torch.stack([yd.sum(dim=(2, 3)), (yd * yd).sum(dim=(2, 3))], dim=-1)
    assert float((sums - ref).abs().max() / ref.abs().max()) < 1e-6
    a_fused, st_fused = ops.raw_gn_forward(y, gw, gb, groups, 1e-6, True, presummed=True)
    torch.cuda.synchronize()
    assert float(ws.abs().max()) == 0.0                                    # the apply pass left the workspace zero again
    y_plain = ops.raw_conv_fprop(x, w5, bias, res, 3, bool(ups), 0, F32, cout, 5)
    a_plain, st_plain = ops.raw_gn_forward(y_plain, gw, gb, groups, 1e-6, True)
    torch.cuda.synchronize()
    assert torch.equal(y, y_plain)
    assert float((a_fused - a_plain).abs().max()) < 2e-5 * float(a_plain.abs().max())
    assert float((st_fused - st_plain).abs().max()) < 1e-5 * float(st_plain.abs().max())


@pytest.mark.parametrize('size', [96, 80])
def test_bf16x3_model_on_sizes_the_split_kernels_only_partly_serve(size):
    """maps whose width is not a multiple of 16 (96 -> 48 -> 24 -> 12; 80 -> 40 -> 20 -> 10) fall back to the exact-fp32 kernels layer
    by layer: a bf16x3 model still steps, and its gradients follow the exact mode's at the split-product level"""
    model_mod = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.model')
    trainer_mod = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.trainer')
    ae = dict(channels=64, num_res_blocks=1, channel_multipliers=(1, 2, 2))
    qc = dict(num_embeddings=128, embedding_dim=64, reinit_every_n_epochs=None, type='standard', params=dict(commitment_cost=0.25))
    tc = dict(lr=1e-4, betas=(0.0, 0.99), eps=1e-8, weight_decay=1e-4, warmup_epochs=None, decay_epochs=None)
    images = torch.rand(2, 3, size, size, generator=torch.Generator().manual_seed(size)).to(DEV)
    grads = {}
    for mode in (torch.float32, 'bf16x3'):
        torch.manual_seed(3)
        m = model_mod.VQVAE(size, ae, qc, None, tc, compute_dtype=mode)
        with torch.no_grad():
            m.quantizer.codebook.weight.mul_(64.0)
        m = m.to(DEV).train()
        tr = trainer_mod.MiniTrainer(num_training_batches=1)
        opt = tr.attach(m)[0]
        opt.zero_grad()
        loss = m.training_step(images, 0)
        loss.backward()
        torch.cuda.synchronize()
        grads[mode] = (float(loss.detach()), opt.flat_g.clone())
    (l0, g0), (l1, g1) = grads[torch.float32], grads['bf16x3']
    assert abs(l0 - l1) < 1e-5 * abs(l0)
    assert float((g0 - g1).norm() / g0.norm()) < 2e-4
    assert not torch.equal(g0, g1)                                # (some layers did take the split-product kernels)


# ---- the 2x2-resampling convs in PHASE form (conv_x3.hip NTAP = 4, layout 6): 4/9 of the multiply-adds, same split products ----
# n, cin, cout, h, w (LOW resolution), bias
PHASE_CASES = [(2, 128, 128, 16, 32, 1), (1, 128, 256, 8, 32, 0), (3, 256, 128, 16, 16, 1), (2, 128, 128, 64, 64, 1), (1, 32, 128, 24, 16, 0)]


@pytest.mark.parametrize('n,cin,cout,h,w,hb', PHASE_CASES)
def test_x3_upsample_conv_phase_form(n, cin, cout, h, w, hb):
    """nearest-x2 + 3x3 conv (autoencoder.py:102-105), forward and data gradient, against fp64 on the un-summed nine taps"""
    g = torch.Generator(device=DEV).manual_seed(cin + cout + h + w + hb)
    x = torch.randn(n, cin, h, w, device=DEV, generator=g).contiguous(memory_format=CL)
    wt = torch.randn(cout, 3, 3, cin, device=DEV, generator=g) / (3 * cin ** 0.5)
    bias = torch.randn(cout, device=DEV, generator=g) if hb else None
    dy = torch.randn(n, cout, 2 * h, 2 * w, device=DEV, generator=g).contiguous(memory_format=CL)
    w4 = ops.pack_weights(wt.reshape(-1), F32, cout, cin, 3, False, 6)
    y = ops.raw_conv_ups_phase(x, w4, bias, cout, False)
    assert y is not None, 'the split-product phase kernel must serve this shape'
    ref = _ref64(x, wt, bias, None, 1)
    torch.cuda.synchronize()
    assert y.shape == ref.shape and y.dtype == F32
    assert float((y.double() - ref).abs().max() / ref.abs().max()) < TOL
    assert float((y.double() - ref).norm() / ref.norm()) < 1e-5
    if cout % 32 == 0 and cin % 128 == 0:
        w4t = ops.pack_weights(wt.reshape(-1), F32, cout, cin, 3, True, 6)
        dx = ops.raw_conv_ups_phase(dy, w4t, None, cin, True)
        assert dx is not None
        xu = F.interpolate(x.double(), scale_factor=2, mode='nearest').requires_grad_(True)
        (dxu,) = torch.autograd.grad(F.conv2d(xu, wt.permute(0, 3, 1, 2).double(), padding=1), xu, dy.double())
        dref = F.avg_pool2d(dxu, 2) * 4.0
        torch.cuda.synchronize()
        assert float((dx.double() - dref).abs().max() / dref.abs().max()) < TOL
        assert float((dx.double() - dref).norm() / dref.norm()) < 1e-5


@pytest.mark.parametrize('n,cin,cout,h,w,gn', [(2, 128, 128, 32, 32, 0), (2, 128, 256, 128, 64, 32), (1, 256, 256, 16, 32, 0), (3, 128, 128, 48, 32, 32)])
def test_x3_pooled_conv_phase_forms(n, cin, cout, h, w, gn):
    """3x3 conv + 2x2 average pool (autoencoder.py:89-91): the forward as a 4x4 stride-2 conv (+ pooled skip, + GroupNorm sums of the
    result) and the data gradient from the POOLED gradient, against fp64; h, w: the conv's (full) resolution"""
    g = torch.Generator(device=DEV).manual_seed(cin + cout + h + gn)
    x = torch.randn(n, cin, h, w, device=DEV, generator=g).contiguous(memory_format=CL)
    wgt = (torch.randn(cout, cin, 3, 3, device=DEV, generator=g) / (3 * cin ** 0.5)).contiguous(memory_format=CL)
    skip_p = torch.randn(n, cout, h // 2, w // 2, device=DEV, generator=g).contiguous(memory_format=CL)
    dyp = torch.randn(n, cout, h // 2, w // 2, device=DEV, generator=g).contiguous(memory_format=CL)
    ops._claim_presummed(x, -1)
    got = ops.raw_conv_pooled_fprop_phase(x, wgt, skip_p, 0.25, gn, x3=True)
    assert got is not None, 'the split-product pooled forward must serve this shape'
    xd = x.double().requires_grad_(True)
    conv = F.conv2d(xd, wgt.double(), padding=1)
    ref = F.avg_pool2d(conv, 2) + skip_p.double()
    torch.cuda.synchronize()
    assert float((got.double() - ref).abs().max() / ref.abs().max()) < TOL
    if gn and (h // 2) * (w // 2) > 1024:
        assert ops.pending_gn() is not None
        cpg = cout // gn
        sums = ops._gn_ws(x.device, n * gn * 2)[:n * gn * 2].clone().view(n, gn, 2)
        gv = got.double().view(n, gn, cpg, -1)
        sref = torch.stack([gv.sum(dim=(2, 3)), (gv * gv).sum(dim=(2, 3))], dim=-1)
        assert float((sums - sref).abs().max() / sref.abs().max()) < 1e-6
        ops._claim_presummed(got, -1)                                # give the workspace back
    assert float(ops._gn_ws(x.device, n * max(gn, 1) * 2).abs().max()) == 0.0
    dx = ops.raw_conv_pooled_dgrad_phase(dyp, wgt, 0.25, x3=True)
    assert dx is not None, 'the split-product pooled data gradient must serve this shape'
    (dref,) = torch.autograd.grad(F.avg_pool2d(conv, 2), xd, dyp.double())
    torch.cuda.synchronize()
    assert dx.shape == dref.shape
    assert float((dx.double() - dref).abs().max() / dref.abs().max()) < TOL
    assert float((dx.double() - dref).norm() / dref.norm()) < 1e-5


def test_x3_phase_forms_inside_the_autograd_functions(monkeypatch):
    """a pooled ResBlock and an Upsample conv in bf16x3 mode: with and without the phase forms the results agree to split-product
    accuracy (same products, 4/9 of them pre-summed in the weights)"""
    n, c, h, w = 2, 128, 64, 64
    g = torch.Generator(device=DEV).manual_seed(11)
    x0 = torch.randn(n, c, h, w, device=DEV, generator=g).contiguous(memory_format=CL)
    mk = lambda *s: torch.randn(*s, device=DEV, generator=g)
    n1w, n1b, n2w, n2b = mk(c), mk(c), mk(c), mk(c)
    c1w = (mk(c, c, 3, 3) / (3 * c ** 0.5)).contiguous(memory_format=CL)
    c2w = (mk(c, c, 3, 3) / (3 * c ** 0.5)).contiguous(memory_format=CL)
    uw = (mk(c, c, 3, 3) / (3 * c ** 0.5)).contiguous(memory_format=CL)
    ub = mk(c)
    dy = mk(n, c, h, w).contiguous(memory_format=CL)
    saved = ops.X3
    outs = []
    try:
        ops.set_conv_products('bf16x3')
        for on in (True, False):
            monkeypatch.setattr(ops, 'X3_PHASE', on)
            ps = [t.clone().requires_grad_(True) for t in (x0, n1w, n1b, c1w, n2w, n2b, c2w, uw, ub)]
            ps[3].data = ps[3].data.contiguous(memory_format=CL); ps[6].data = ps[6].data.contiguous(memory_format=CL)
            ps[7].data = ps[7].data.contiguous(memory_format=CL)
            y = ops.res_block(ps[0], ps[1], ps[2], ps[3], ps[4], ps[5], ps[6], None, 32, 1e-6, pool=True)      # -> h/2
            y = ops.conv2d(y, ps[7], ps[8], ups=True)                                                          # -> h
            grads = torch.autograd.grad(y, ps, dy)
            torch.cuda.synchronize()
            outs.append([y.detach()] + [t.detach() for t in grads])
    finally:
        ops.X3 = saved
    for a, b in zip(*outs):
        assert float((a - b).abs().max()) / float(b.abs().max()) < 1e-4


@pytest.mark.parametrize('phase', [1, 0])
@pytest.mark.parametrize('n,cin,cout,h,w', [(2, 128, 128, 16, 32), (1, 64, 256, 8, 8), (3, 256, 64, 24, 16), (2, 128, 128, 64, 64), (5, 64, 64, 8, 24)])
def test_x3_upsample_conv_weight_gradient_phase_form(n, cin, cout, h, w, phase):
    """dW of nearest-x2 + 3x3 conv (autoencoder.py:102-105): four 2x2-window phases on the low-resolution grid (tuning slot
    X3_WGRAD_PHASE = 1, the default) and the tap form (0) against fp64; the target ACCUMULATES"""
    g = torch.Generator(device=DEV).manual_seed(n + cin + cout + h)
    x = torch.randn(n, cin, h, w, device=DEV, generator=g).contiguous(memory_format=CL)
    dy = torch.randn(n, cout, 2 * h, 2 * w, device=DEV, generator=g).contiguous(memory_format=CL)
    pre = torch.randn(cout, 3, 3, cin, device=DEV, generator=g).permute(0, 3, 1, 2)
    dw = pre.clone(memory_format=torch.preserve_format)
    lib = native.lib()
    try:
        native.check(lib.vqk_set_tuning(b'X3_WGRAD_PHASE', phase), 'set_tuning')
        ops.raw_conv_wgrad(x, dy, 3, True, out=dw, x3=True)
    finally:
        native.check(lib.vqk_set_tuning(b'X3_WGRAD_PHASE', 1), 'set_tuning')
    xu = F.interpolate(x.double(), scale_factor=2, mode='nearest')
    wd = torch.zeros(cout, cin, 3, 3, device=DEV, dtype=torch.float64, requires_grad=True)
    (F.conv2d(xu, wd, None, padding=1) * dy.double()).sum().backward()
    ref = wd.grad + pre.double()
    torch.cuda.synchronize()
    assert float((dw.double() - ref).abs().max()) / float(ref.abs().max()) < TOL
    assert float((dw.double() - ref).norm() / ref.norm()) < 1e-5


@pytest.mark.parametrize('n,cin,cout,h,w', [(2, 128, 128, 32, 32), (1, 64, 256, 16, 48), (3, 256, 64, 48, 16), (2, 128, 128, 128, 128)])
def test_x3_pooled_conv_weight_gradient_phase_form(n, cin, cout, h, w):
    """dW of 3x3 conv + 2x2 average pool (autoencoder.py:89-91) from the POOLED gradient: the 16 taps of the 4x4 stride-2 window at pooled
    resolution folded onto the 3x3 taps (vqk_conv2d_wgrad_x3_f32, ups = 2), against fp64 autograd; h, w: the conv's resolution"""
    g = torch.Generator(device=DEV).manual_seed(n + cin + cout + w)
    x = torch.randn(n, cin, h, w, device=DEV, generator=g).contiguous(memory_format=CL)
    dyp = torch.randn(n, cout, h // 2, w // 2, device=DEV, generator=g).contiguous(memory_format=CL)
    pre = torch.randn(cout, 3, 3, cin, device=DEV, generator=g).permute(0, 3, 1, 2)
    dw = pre.clone(memory_format=torch.preserve_format)
    assert ops.raw_conv_wgrad_pooled_x3(x, dyp, 0.25, dw), 'the pooled phase form must serve this shape'
    wd = torch.zeros(cout, cin, 3, 3, device=DEV, dtype=torch.float64, requires_grad=True)
    (F.avg_pool2d(F.conv2d(x.double(), wd, None, padding=1), 2) * dyp.double()).sum().backward()
    ref = wd.grad + pre.double()
    torch.cuda.synchronize()
    assert float((dw.double() - ref).abs().max()) / float(ref.abs().max()) < TOL
    assert float((dw.double() - ref).norm() / ref.norm()) < 1e-5
    # not served: odd pooled sizes fall back (nothing launched, target untouched)
    x2 = torch.randn(1, 64, 24, 24, device=DEV, generator=g).contiguous(memory_format=CL)
    d2 = torch.randn(1, 64, 12, 12, device=DEV, generator=g).contiguous(memory_format=CL)
    t2 = torch.zeros(64, 3, 3, 64, device=DEV).permute(0, 3, 1, 2)
    assert not ops.raw_conv_wgrad_pooled_x3(x2, d2, 0.25, t2) and float(t2.abs().max()) == 0.0
