"""VQ-GAN loss-path operators over the vqk C-ABI: the reference's two plugins (``bias_act``, ``upfirdn2d``) with their call
surface, conv + activation with the StyleGAN2 gains (``ConvActFn`` and its double-differentiable pieces), NHWC resampling, pooling,
the LPIPS taps, minibatch stddev, the fused discriminator block and the losses.  Private part of :mod:`ops` (imported at the end of
``ops.py``, which re-exports every name); shared infrastructure and the switches are reached through ``core``."""
from __future__ import annotations

import torch

from . import _native
from . import ops as core

# ------------------------------------------------------------------------------------------------------
# StyleGAN2 plugin ops (same call surface as the reference's python wrappers)
# ------------------------------------------------------------------------------------------------------
_ACT_IDX = {'linear': 1, 'lrelu': 3}


def _bias_act_raw(x, b, yref, dy, grad, dim, act, alpha, gain, clamp):
    x = x.contiguous()
    y = torch.empty_like(x)
    inner = 1
    for s in x.shape[dim + 1:]:
        inner *= s
    st = _native.lib().vqk_bias_act(x.data_ptr(), core._p(b), 0, core._p(yref), core._p(dy), y.data_ptr(), x.numel(), inner,
                                    x.shape[dim] if b is not None else 1, grad, _ACT_IDX[act], alpha, gain, clamp,
                                    core._stream())
    _native.check(st, 'bias_act')
    return y


class BiasActFn(torch.autograd.Function):
    """bias_act.py:129-210 (lrelu / linear, first-order; second order re-applies the same mask)."""

    @staticmethod
    def forward(ctx, x, b, dim, act, alpha, gain, clamp):
        core._require_gpu(x)
        y = _bias_act_raw(x, b, None, None, 0, dim, act, alpha, gain, clamp)
        ctx.save_for_backward(y)
        ctx.cfg = (dim, act, alpha, gain, clamp, b is not None, x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dim, act, alpha, gain, clamp, has_b, shape = ctx.cfg
        dx = BiasActGradFn.apply(dy.contiguous(), y, dim, act, alpha, gain, clamp)
        db = None
        if has_b:
            db = dx.sum([i for i in range(dx.ndim) if i != dim])
        return dx, db, None, None, None, None, None


class BiasActGradFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, y, dim, act, alpha, gain, clamp):
        dx = _bias_act_raw(dy, None, y, None, 1, dim, act, alpha, gain, clamp) if act != 'linear' or gain != 1 or clamp >= 0 \
            else dy
        ctx.save_for_backward(y)
        ctx.cfg = (dim, act, alpha, gain, clamp)
        return dx

    @staticmethod
    def backward(ctx, d_dx):
        (y,) = ctx.saved_tensors
        dim, act, alpha, gain, clamp = ctx.cfg
        return BiasActGradFn.apply(d_dx.contiguous(), y, dim, act, alpha, gain, clamp), None, None, None, None, None, None


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """Same signature and defaults as the reference's ``bias_act`` (bias_act.py:55-89)."""
    defaults = {'linear': (0.0, 1.0), 'lrelu': (0.2, 2.0 ** 0.5)}
    if act not in defaults:
        raise RuntimeError(f'vqk: bias_act activation {act!r} is not on the discriminator path')
    alpha = float(defaults[act][0] if alpha is None else alpha)
    gain = float(defaults[act][1] if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    return BiasActFn.apply(x, b, dim, act, alpha, gain, clamp)


def _upfirdn2d_raw(x, f, up, down, pad, flip, gain):
    x = x.contiguous()
    n, c, h, w = x.shape
    fh, fw = f.shape
    upx, upy = up
    downx, downy = down
    px0, px1, py0, py1 = pad
    ow = (w * upx + px0 + px1 - fw + downx) // downx
    oh = (h * upy + py0 + py1 - fh + downy) // downy
    y = torch.empty((n, c, oh, ow), dtype=x.dtype, device=x.device)
    st = _native.lib().vqk_upfirdn2d(x.data_ptr(), f.contiguous().data_ptr(), y.data_ptr(), n, c, h, w, fh, fw, upx, upy,
                                     downx, downy, px0, px1, py0, py1, int(flip), gain, oh, ow, core._stream())
    _native.check(st, 'upfirdn2d')
    return y


class Upfirdn2dFn(torch.autograd.Function):
    """upfirdn2d.py:214-268: linear op, backward = the same op with up<->down, flipped filter."""

    @staticmethod
    def forward(ctx, x, f, up, down, pad, flip, gain):
        core._require_gpu(x)
        ctx.save_for_backward(f)
        ctx.cfg = (up, down, pad, flip, gain, x.shape)
        return _upfirdn2d_raw(x, f, up, down, pad, flip, gain)

    @staticmethod
    def backward(ctx, dy):
        (f,) = ctx.saved_tensors
        up, down, pad, flip, gain, xs = ctx.cfg
        fh, fw = f.shape
        _, _, ih, iw = xs
        _, _, oh, ow = dy.shape
        p = (fw - pad[0] - 1, iw * up[0] - ow * down[0] + pad[0] - up[0] + 1,
             fh - pad[2] - 1, ih * up[1] - oh * down[1] + pad[2] - up[1] + 1)
        return Upfirdn2dFn.apply(dy, f, down, up, p, not flip, gain), None, None, None, None, None, None


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    """Same signature as the reference's ``upfirdn2d`` (upfirdn2d.py:120-164); 2-D filters only."""
    up = (up, up) if isinstance(up, int) else tuple(up)
    down = (down, down) if isinstance(down, int) else tuple(down)
    if isinstance(padding, int):
        padding = (padding,) * 4
    padding = tuple(padding)
    if len(padding) == 2:
        padding = (padding[0], padding[0], padding[1], padding[1])
    if f.ndim != 2:
        raise RuntimeError('vqk: upfirdn2d expects a 2-D FIR filter')
    return Upfirdn2dFn.apply(x, f.to(torch.float32), up, down, padding, bool(flip_filter), float(gain))


# ------------------------------------------------------------------------------------------------------
# VQ-GAN loss path: general conv + activation, NHWC resampling, pooling, LPIPS tap, minibatch-stddev, losses
# ------------------------------------------------------------------------------------------------------


def _conv_general_raw(x, wq, bias, residual, cout, k, stride, pad, mode, h_out, w_out, act, acc_scale, out_gain, out_dtype,
                      wlayout=0):
    n, cin, h, w = x.shape
    y = core.empty_nhwc(n, cout, h_out, w_out, out_dtype, x.device)
    flops = 2.0 * n * h_out * w_out * cout * cin * k * k
    st = core._timed(core._fprop_kernel_name(x.dtype, wlayout, (n, h_out, w_out, cin, cout, act, out_dtype) if stride == 1 and mode != 2 else None), flops,
                lambda: _native.lib().vqk_conv2d_general(core.dcode(x.dtype), x.data_ptr(), wq.data_ptr(), core._p(bias), core._p(residual),
                                                         y.data_ptr(), core.dcode(out_dtype), n, h, w, cin, cout, k, stride, pad,
                                                         mode, h_out, w_out, act, float(acc_scale), float(out_gain), wlayout,
                                                         core.zero_page(x.device).data_ptr(), core._stream()))
    _native.check(st, 'conv2d_general')
    return y


def _s2_served(dt, out_dtype, n, h, w, h_out, w_out, cin, cout, k, stride, pad, backward: bool) -> bool:
    """the stride-2 3x3 conv without padding on a (2 h_out + 1) x (2 w_out + 1) input (the discriminator's down-sampling conv
    after its blur, discriminator.py:95 / conv2d_resample.py:119-122) has a matrix/auxiliary-wave form (vqk_conv2d_s2_*)"""
    if not (k == 3 and stride == 2 and pad == 0 and dt == torch.bfloat16 and out_dtype == torch.bfloat16
            and h == 2 * h_out + 1 and w == 2 * w_out + 1):
        return False
    return bool(_native.lib().vqk_conv2d_s2_supported(core.dcode(dt), n, h_out, w_out, cin, cout, int(backward)))


def _conv_s2_fprop_raw(x, wq, bias, cout, h_out, w_out, act, acc_scale, out_gain):
    n, cin, h, w = x.shape
    y = core.empty_nhwc(n, cout, h_out, w_out, x.dtype, x.device)
    flops = 2.0 * n * h_out * w_out * cout * cin * 9
    st = core._timed('conv3x3_mx_kernel<bf16> (stride 2)', flops,
                lambda: _native.lib().vqk_conv2d_s2_fprop(core.dcode(x.dtype), x.data_ptr(), wq.data_ptr(), core._p(bias), y.data_ptr(), n,
                                                          h_out, w_out, cin, cout, act, float(acc_scale), float(out_gain),
                                                          core.zero_page(x.device).data_ptr(), core._stream()))
    _native.check(st, 'conv2d_s2_fprop')
    return y


def _packed_w4(weight, w4, cin, cout_pad, dt, k, transpose, layout):
    """cached operand of a conv parameter; ``w4``: its [O,I,k,k] view (2-D fully connected weights are 1x1 convs)"""
    return core.packed_weight(weight, cin, cout_pad, dt, k, transpose, layout, shape4=tuple(w4.shape))


class ConvActFn(torch.autograd.Function):
    """y = out_gain * act(conv(x, W) * wgain + bias), stride in {1,2}, explicit zero padding.

    The StyleGAN2 ``Conv2dLayer`` / ``FullyConnectedLayer`` arithmetic (discriminator.py:104-120, :164-173: runtime
    weight gain 1/sqrt(fan_in), fused bias + activation + gain = the reference's ``bias_act`` plugin) and the VGG16
    conv+bias+ReLU of LPIPS, all in the conv kernel's epilogue.  Backward: t = out_gain * act'(y) * dy (``bias_act``
    grad=1), db = colsum(t), dx = wgain * dgrad(t) (zero-stuffed gather for stride 2), dW = wgain * wgrad(x, t)."""

    @staticmethod
    def forward(ctx, x, weight, bias, k: int, stride: int, pad: int, act: int, wgain: float, out_gain: float, out_dtype):
        core._require_gpu(x)
        x = core.nhwc(x)
        dt = x.dtype
        out_dtype = out_dtype or dt
        o, i = weight.shape[0], weight.shape[1]
        cin = x.shape[1]
        e = max(core.epc(dt), core.epc(out_dtype))
        cout_pad = -(-o // e) * e
        if cin < i or cin % core.epc(dt):
            raise RuntimeError(f'vqk: conv input has {cin} channels, weight expects {i}')
        n, _, h, w = x.shape
        h_out = (h + 2 * pad - k) // stride + 1
        w_out = (w + 2 * pad - k) // stride + 1
        plain = stride == 1 and pad == k // 2
        s2 = _s2_served(dt, out_dtype, n, h, w, h_out, w_out, cin, cout_pad, k, stride, pad, False)
        layout = core.weight_layout(dt, n, h, w, cin, cout_pad, k, False, out_dtype) if plain else (1 if s2 else 0)
        w4 = weight.reshape(o, i, k, k)
        wq = _packed_w4(weight, w4, cin, cout_pad, dt, k, False, layout)
        b32 = core.padded_vector(bias, cout_pad) if bias is not None else None
        if (k == 1 and plain and cin == 8 and dt == torch.bfloat16 and out_dtype == dt and w % 32 == 0 and cout_pad % 8 == 0):
            # a 1x1 conv on the padded 3-channel image (the discriminator's fromrgb, discriminator.py:198-199): the thin-input
            # 3x3 kernel with the weights at the centre tap (zero elsewhere) writes its 2*Cout bytes per pixel at memory speed;
            # the im2col kernel took 264 us for 8 -> 128 @256^2, bs 16
            wq3 = core.packed_weight(weight, 8, cout_pad, dt, 3, False, 0, shape4=(o, i, 1, 1), kind='centre3')
            y = _conv_general_raw(x, wq3, b32, None, cout_pad, 3, 1, 1, 0, h_out, w_out, act, wgain, out_gain, out_dtype, 0)
        elif s2:
            y = _conv_s2_fprop_raw(x, wq, b32, cout_pad, h_out, w_out, act, wgain, out_gain)
        else:
            y = _conv_general_raw(x, wq, b32, None, cout_pad, k, stride, pad, 0, h_out, w_out, act, wgain, out_gain, out_dtype,
                                  layout)
        ctx.save_for_backward(x, y)
        ctx.refs = (weight, bias)
        ctx.cfg = (k, stride, pad, act, wgain, out_gain, o, i, cin, cout_pad, dt)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        dx, dw, db = _conv_act_backward(x, y, ctx.refs, ctx.cfg, ctx.needs_input_grad, dy)
        return dx, dw, db, None, None, None, None, None, None, None


def _conv_act_backward(x, y, refs, cfg, needs, dy, make_t=None, dx_residual=None):
    """ConvActFn's backward as a plain function (DiscBlockFn composes three of them).  ``make_t(scale, dbsum, db_scale)``: the caller
    produces t = scale * act'(y) * dy itself (fused into the pass that produces dy) and adds db_scale * t's column sums to ``dbsum``
    when that is not None; ``dx_residual``: added to the data gradient in the conv kernel's epilogue."""
    weight, bias = refs
    k, stride, pad, act, wgain, out_gain, o, i, cin, cout_pad, dt = cfg
    lib, st = _native.lib(), core._stream()
    # t and dx are built from differentiable Functions so that R1 (autograd.grad(..., create_graph=True) through
    # this backward, loss.py:98-112) can differentiate them again; in an ordinary backward they record nothing.
    param_grads = core._grad_modes()[1]
    want_db = bias is not None and needs[2] and param_grads
    want_dw = needs[1] and param_grads
    dyn = core.nhwc(dy) if dy is not None else None
    dy_dtype = dyn.dtype if dyn is not None else dt
    dbsum = db_tgt = None
    gscale = 1.0
    # weight gradient straight into the optimizer's flat gradient arena (unpadded layers of a FlatAdamW-owned module): the
    # 1/sqrt(fan_in) weight gain then rides in t (t' = wgain * t: dx = dgrad(t', W), dW += wgrad(x, t'), db = colsum(t') / wgain)
    # -- no zero-filled temporary, no scale pass, no accumulate pass per parameter
    tgt = tgt_lin = None
    if (want_dw and dy_dtype == dt and cout_pad == o and cin == i and wgain > 0.0
            and not torch.is_grad_enabled()):            # (a create_graph pass -- R1's inner autograd.grad -- must not touch .grad)
        if act != 0:
            tgt = core.direct_grad(weight)
        elif make_t is None and core.DIRECT_LINEAR_WGRAD:
            # linear layer (skip convs, the last fully connected layer): t = dy is no pass, the gains ride in the weight-gradient
            # kernel's own scale (vqk_conv2d_wgrad_general_scaled) -- also straight into the arena
            tgt_lin = core.direct_grad(weight)
    fold = float(wgain) if tgt is not None else 1.0
    if make_t is not None:
        db_scale = 1.0
        if want_db:
            if core.DIRECT_BIAS_GRAD and cout_pad == o and not torch.is_grad_enabled():
                db_tgt = core.direct_grad(bias)               # the caller's column-sum pass adds (1 / fold) * colsum(t) to the arena
            if db_tgt is not None:
                dbsum, db_scale = db_tgt, 1.0 / fold
            else:
                dbsum = torch.zeros(cout_pad, dtype=torch.float32, device=x.device)
        t = make_t(float(out_gain) * fold, dbsum, db_scale)
    elif act == 0 and dyn.dtype == dt:
        # linear layer (the discriminator's skip convs, its last fully connected layer): t = out_gain * dy is no pass over
        # the tensor -- the scalar rides in the data- / weight-gradient scale (and on the bias sum)
        t, gscale = dyn, float(out_gain)
    else:
        if want_db and dyn.dtype == dt:                  # the bias gradient rides in the act-backward pass
            vec = 4 if dt == torch.float32 else 8
            if core.DIRECT_BIAS_GRAD and cout_pad == o and cout_pad % vec == 0 and cout_pad // vec <= 256 and not torch.is_grad_enabled():
                db_tgt = core.direct_grad(bias)               # ... straight into the optimizer's arena (no zero fill, scale, add)
            dbsum = db_tgt if db_tgt is not None else torch.zeros(cout_pad, dtype=torch.float32, device=x.device)
        t = ActBwdFn.apply(dyn, y, act, float(out_gain) * fold, dbsum, 1.0 / fold if db_tgt is not None else 1.0)
    tc = t if t.dtype == dt else core.nhwc(t.to(dt))
    n, _, h, w = x.shape
    _, _, h_out, w_out = tc.shape
    dx = dw = db = None
    if needs[0]:
        dx = ConvDgradFn.apply(tc, weight, k, stride, pad, 1.0 if tgt is not None else float(wgain) * gscale, cin, cout_pad, h, w,
                               dx_residual)
    if tgt is not None:
        _native.check(lib.vqk_conv2d_wgrad_general(core.dcode(dt), x.data_ptr(), tc.detach().data_ptr(), tgt.data_ptr(), n, h, w, cin,
                                                   cout_pad, k, stride, pad, 0, h_out, w_out,
                                                   core.zero_page(x.device).data_ptr(), st), 'conv2d_wgrad_general')
    elif tgt_lin is not None:
        _native.check(lib.vqk_conv2d_wgrad_general_scaled(core.dcode(dt), x.data_ptr(), tc.detach().data_ptr(), tgt_lin.data_ptr(), n, h, w,
                                                          cin, cout_pad, k, stride, pad, 0, h_out, w_out, float(wgain) * gscale,
                                                          core.zero_page(x.device).data_ptr(), st), 'conv2d_wgrad_general')
    elif want_dw:
        tcd = tc.detach()
        dwp = torch.zeros((cout_pad, k, k, cin), dtype=torch.float32, device=x.device)
        _native.check(lib.vqk_conv2d_wgrad_general(core.dcode(dt), x.data_ptr(), tcd.data_ptr(), dwp.data_ptr(), n, h, w, cin,
                                                   cout_pad, k, stride, pad, 0, h_out, w_out,
                                                   core.zero_page(x.device).data_ptr(), st), 'conv2d_wgrad_general')
        if wgain * gscale != 1.0:
            _native.check(lib.vqk_axpby(core.F32, dwp.data_ptr(), 0, dwp.data_ptr(), float(wgain) * gscale, 0.0, dwp.numel(), st), 'axpby')
        dw = dwp.permute(0, 3, 1, 2)[:o, :i].reshape(weight.shape)
    if want_db and db_tgt is None:
        fused = dbsum is not None and (make_t is not None or (cout_pad % (4 if dt == torch.float32 else 8) == 0
                                                               and cout_pad // (4 if dt == torch.float32 else 8) <= 256))
        db = (dbsum if fused else core.raw_colsum(n * h_out * w_out, cout_pad, tc.detach()))[:o]
        if gscale != 1.0 or fold != 1.0:
            db = db * (gscale / fold)
    return dx, dw, db


class ActBwdFn(torch.autograd.Function):
    """t = scale * act'(y) * dy -- linear in dy for the piecewise-linear / saved-output activations used here, so
    its own backward is the same op (bias_act.py:197-198: lrelu / relu have no second-order term)"""

    @staticmethod
    def forward(ctx, dy, y, act: int, scale: float, colsum=None, colsum_scale: float = 1.0):
        """colsum: fp32 [C] buffer that ALSO receives colsum_scale * the column sums of the result (the conv's bias gradient) -- one pass"""
        t = torch.empty_like(dy, memory_format=core._CL)
        n, c, h, w = dy.shape
        v = 4 if dy.dtype == torch.float32 else 8
        if colsum is not None and c % v == 0 and c // v <= 256:
            _native.check(_native.lib().vqk_act_backward_colsum_scaled(core.dcode(dy.dtype), dy.data_ptr(), y.data_ptr(), t.data_ptr(),
                                                                       n * h * w, c, act, scale, float(colsum_scale),
                                                                       colsum.data_ptr(), core._stream()), 'act_backward_colsum')
            ctx.fused_colsum = True
        else:
            _native.check(_native.lib().vqk_act_backward(core.dcode(dy.dtype), dy.data_ptr(), y.data_ptr(), t.data_ptr(), dy.numel(),
                                                         act, scale, core._stream()), 'act_backward')
            ctx.fused_colsum = False
        ctx.save_for_backward(y)
        ctx.cfg = (act, scale)
        if act == 1:
            ctx.set_materialize_grads(False)
        return t

    @staticmethod
    def backward(ctx, v):
        (y,) = ctx.saved_tensors
        act, scale = ctx.cfg
        if act == 1:
            raise NotImplementedError('second-order tanh epilogue is not on any path')
        return ActBwdFn.apply(core.nhwc(v), y, act, scale), None, None, None, None, None


class ConvDgradFn(torch.autograd.Function):
    """dx = wgain * dgrad(t, W): bilinear in (t, W).  Differentiating it (R1) gives a FORWARD conv of the incoming
    cotangent and a wgrad with the cotangent in the role of the layer input."""

    @staticmethod
    def forward(ctx, t, weight, k: int, stride: int, pad: int, wgain: float, cin: int, cout_pad: int, h: int, w: int,
                residual=None):
        """residual (a constant of the differentiation: DiscBlockFn's other branch gradient) is added in the kernel's epilogue"""
        dt = t.dtype
        o, i = weight.shape[0], weight.shape[1]
        w4 = weight.detach().reshape(o, i, k, k)
        n, _, h_out, w_out = t.shape
        if stride == 1 and pad == k // 2:
            layout = core.weight_layout(dt, n, h_out, w_out, cout_pad, cin, k, False)
            wt = _packed_w4(weight, w4, cin, cout_pad, dt, k, True, layout)
            dx = _conv_general_raw(t, wt, None, residual, cin, k, 1, k // 2, 0, h, w, 0, wgain, 1.0, dt, layout)
            residual = None
        elif _s2_served(dt, dt, n, h, w, h_out, w_out, cin, cout_pad, k, stride, pad, True):
            wt = _packed_w4(weight, w4, cin, cout_pad, dt, k, True, 0)
            w3 = _packed_w4(weight, w4, cin, cout_pad, dt, k, True, 3)
            dx = core.empty_nhwc(n, cin, h, w, dt, t.device)
            st = core._timed('conv3x3_mx_kernel<bf16> (stride-2 dgrad phases)', 2.0 * n * h_out * w_out * cout_pad * cin * 9,
                        lambda: _native.lib().vqk_conv2d_s2_dgrad(core.dcode(dt), t.data_ptr(), w3.data_ptr(), wt.data_ptr(), dx.data_ptr(),
                                                                  n, h_out, w_out, cin, cout_pad, float(wgain),
                                                                  core.zero_page(t.device).data_ptr(), core._stream()))
            _native.check(st, 'conv2d_s2_dgrad')
        else:
            wt = _packed_w4(weight, w4, cin, cout_pad, dt, k, True, 0)
            dx = _conv_general_raw(t, wt, None, None, cin, k, 1, k - 1 - pad, 2 if stride == 2 else 0, h, w, 0, wgain, 1.0,
                                   dt, 0)
        if residual is not None:                         # (the strided / padded forms have no residual operand)
            dx = AddFn.apply(dx, residual)
        ctx.save_for_backward(t)
        ctx.refs = (weight,)
        ctx.cfg = (k, stride, pad, wgain, cin, cout_pad, o, i)
        return dx

    @staticmethod
    def backward(ctx, v):
        (t,) = ctx.saved_tensors
        (weight,) = ctx.refs
        k, stride, pad, wgain, cin, cout_pad, o, i = ctx.cfg
        v = core.nhwc(v)
        dt = t.dtype
        n, _, h, w = v.shape
        _, _, h_out, w_out = t.shape
        w4 = weight.detach().reshape(o, i, k, k)
        lib, st = _native.lib(), core._stream()
        d_t = d_w = None
        if ctx.needs_input_grad[0] and _s2_served(dt, dt, n, h, w, h_out, w_out, cin, cout_pad, k, stride, pad, False):
            d_t = _conv_s2_fprop_raw(v, _packed_w4(weight, w4, cin, cout_pad, dt, k, False, 1), None, cout_pad, h_out, w_out, 0,
                                     wgain, 1.0)
        elif ctx.needs_input_grad[0]:
            wq = _packed_w4(weight, w4, cin, cout_pad, dt, k, False, 0)
            d_t = _conv_general_raw(v, wq, None, None, cout_pad, k, stride, pad, 0, h_out, w_out, 0, wgain, 1.0, dt, 0)
        if ctx.needs_input_grad[1]:
            dwp = torch.zeros((cout_pad, k, k, cin), dtype=torch.float32, device=v.device)
            _native.check(lib.vqk_conv2d_wgrad_general(core.dcode(dt), v.data_ptr(), t.data_ptr(), dwp.data_ptr(), n, h, w, cin,
                                                       cout_pad, k, stride, pad, 0, h_out, w_out,
                                                       core.zero_page(v.device).data_ptr(), st), 'conv2d_wgrad_general')
            if wgain != 1.0:
                _native.check(lib.vqk_axpby(core.F32, dwp.data_ptr(), 0, dwp.data_ptr(), float(wgain), 0.0, dwp.numel(), st), 'axpby')
            d_w = dwp.permute(0, 3, 1, 2)[:o, :i].reshape(weight.shape)
        return d_t, d_w, None, None, None, None, None, None, None, None, None


def conv_act(x, weight, bias=None, k=3, stride=1, pad=None, act='linear', wgain=1.0, out_gain=1.0, out_dtype=None):
    return ConvActFn.apply(x, weight, bias, k, stride, k // 2 if pad is None else pad, core.ACT_CODE[act], float(wgain),
                           float(out_gain), out_dtype)


class UpfirdnNhwcFn(torch.autograd.Function):
    """upfirdn2d on NHWC activations (same op, same padding algebra, same backward rule as upfirdn2d.py:214-268)."""

    @staticmethod
    def forward(ctx, x, f, up, down, pad, flip, gain):
        core._require_gpu(x)
        x = core.nhwc(x)
        n, c, h, w = x.shape
        fh, fw = f.shape
        ow = (w * up + pad[0] + pad[1] - fw + down) // down
        oh = (h * up + pad[2] + pad[3] - fh + down) // down
        y = core.empty_nhwc(n, c, oh, ow, x.dtype, x.device)
        st = _native.lib().vqk_upfirdn2d_nhwc(core.dcode(x.dtype), x.data_ptr(), f.data_ptr(), y.data_ptr(), n, h, w, c, fh, fw,
                                              up, up, down, down, pad[0], pad[1], pad[2], pad[3], int(flip), float(gain),
                                              oh, ow, core._stream())
        _native.check(st, 'upfirdn2d_nhwc')
        ctx.save_for_backward(f)
        ctx.cfg = (up, down, pad, flip, gain, (h, w))
        return y

    @staticmethod
    def backward(ctx, dy):
        (f,) = ctx.saved_tensors
        up, down, pad, flip, gain, (ih, iw) = ctx.cfg
        fh, fw = f.shape
        _, _, oh, ow = dy.shape
        p = (fw - pad[0] - 1, iw * up - ow * down + pad[0] - up + 1, fh - pad[2] - 1, ih * up - oh * down + pad[2] - up + 1)
        return UpfirdnNhwcFn.apply(dy, f, down, up, p, not flip, gain), None, None, None, None, None, None


def upfirdn2d_nhwc(x, f, up=1, down=1, padding=(0, 0, 0, 0), flip_filter=False, gain=1.0):
    return UpfirdnNhwcFn.apply(x, f.to(torch.float32).contiguous(), int(up), int(down), tuple(padding), bool(flip_filter),
                               float(gain))


class MaxPool2x2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        core._require_gpu(x)
        x = core.nhwc(x)
        n, c, h, w = x.shape
        y = core.empty_nhwc(n, c, h // 2, w // 2, x.dtype, x.device)
        _native.check(_native.lib().vqk_maxpool2x2(core.dcode(x.dtype), x.data_ptr(), 0, y.data_ptr(), n, h, w, c, 0, core._stream()), 'maxpool')
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        n, c, h, w = x.shape
        dx = torch.empty_like(x, memory_format=core._CL)
        _native.check(_native.lib().vqk_maxpool2x2(core.dcode(x.dtype), x.data_ptr(), core.nhwc(dy).data_ptr(), dx.data_ptr(), n, h, w, c,
                                                   1, core._stream()), 'maxpool_backward')
        return dx


class ChannelAffineFn(torch.autograd.Function):
    """y = x * scale[c] + shift[c] (constants)"""

    @staticmethod
    def forward(ctx, x, scale, shift):
        core._require_gpu(x)
        x = core.nhwc(x)
        n, c, h, w = x.shape
        y = torch.empty_like(x, memory_format=core._CL)
        _native.check(_native.lib().vqk_channel_affine(core.dcode(x.dtype), x.data_ptr(), scale.data_ptr(), core._p(shift), y.data_ptr(),
                                                       n * h * w, c, core._stream()), 'channel_affine')
        ctx.save_for_backward(scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        (scale,) = ctx.saved_tensors
        dy = core.nhwc(dy)
        n, c, h, w = dy.shape
        dx = torch.empty_like(dy, memory_format=core._CL)
        _native.check(_native.lib().vqk_channel_affine(core.dcode(dy.dtype), dy.data_ptr(), scale.data_ptr(), 0, dx.data_ptr(),
                                                       n * h * w, c, core._stream()), 'channel_affine')
        return dx, None, None


class LpipsTapFn(torch.autograd.Function):
    """per-image LPIPS contribution of one feature tap; gradient flows to ``fy`` (the reconstruction branch) only"""

    @staticmethod
    def forward(ctx, fx, fy, lin):
        core._require_gpu(fx)
        fx, fy = core.nhwc(fx), core.nhwc(fy)
        n, c, h, w = fx.shape
        out = torch.zeros(n, dtype=torch.float32, device=fx.device)
        _native.check(_native.lib().vqk_lpips_tap(core.dcode(fx.dtype), fx.data_ptr(), fy.data_ptr(), lin.data_ptr(), n, h * w, c,
                                                  out.data_ptr(), 0, 1.0, 0, core._stream()), 'lpips_tap')
        ctx.save_for_backward(fx, fy, lin)
        return out

    @staticmethod
    def backward(ctx, dout):
        fx, fy, lin = ctx.saved_tensors
        n, c, h, w = fx.shape
        # dout is [n]: the kernel scales every image by its own upstream gradient (no host-side test of the values, which
        # would be a device -> host sync in the middle of the backward)
        dfy = torch.empty_like(fy, memory_format=core._CL)
        _native.check(_native.lib().vqk_lpips_tap(core.dcode(fx.dtype), fx.data_ptr(), fy.data_ptr(), lin.data_ptr(), n, h * w, c,
                                                  0, dout.contiguous().float().data_ptr(), 1.0, dfy.data_ptr(), core._stream()),
                      'lpips_tap_backward')
        return None, dfy, None


class LpipsTapsFn(torch.autograd.Function):
    """sum over the feature taps of the per-image LPIPS contributions (lpips.py: the five taps' terms added up) as ONE node: the
    tap kernels accumulate into one zero-filled [B] vector -- no fill and no add launch per tap; gradients to the ``fy`` only"""

    @staticmethod
    def forward(ctx, ntap: int, *args):
        fxs, fys, lins = args[:ntap], args[ntap:2 * ntap], args[2 * ntap:3 * ntap]
        core._require_gpu(fxs[0])
        fxs, fys = [core.nhwc(t) for t in fxs], [core.nhwc(t) for t in fys]
        n = fxs[0].shape[0]
        out = torch.zeros(n, dtype=torch.float32, device=fxs[0].device)
        lib, st = _native.lib(), core._stream()
        for fx, fy, lin in zip(fxs, fys, lins):
            _, c, h, w = fx.shape
            _native.check(lib.vqk_lpips_tap(core.dcode(fx.dtype), fx.data_ptr(), fy.data_ptr(), lin.data_ptr(), n, h * w, c,
                                            out.data_ptr(), 0, 1.0, 0, st), 'lpips_tap')
        ctx.save_for_backward(*fxs, *fys, *lins)
        ctx.ntap = ntap
        return out

    @staticmethod
    def backward(ctx, dout):
        k = ctx.ntap
        saved = ctx.saved_tensors
        fxs, fys, lins = saved[:k], saved[k:2 * k], saved[2 * k:]
        d = dout.contiguous().float()
        lib, st = _native.lib(), core._stream()
        grads = []
        for fx, fy, lin in zip(fxs, fys, lins):
            n, c, h, w = fx.shape
            dfy = torch.empty_like(fy, memory_format=core._CL)
            _native.check(lib.vqk_lpips_tap(core.dcode(fx.dtype), fx.data_ptr(), fy.data_ptr(), lin.data_ptr(), n, h * w, c,
                                            0, d.data_ptr(), 1.0, dfy.data_ptr(), st), 'lpips_tap_backward')
            grads.append(dfy)
        return (None,) + (None,) * k + tuple(grads) + (None,) * k


class MbstdFn(torch.autograd.Function):
    """minibatch-stddev feature appended as one extra channel (discriminator.py:277-293); output channels are
    padded with zeros to a whole 16-byte chunk"""

    @staticmethod
    def forward(ctx, x, group: int):
        core._require_gpu(x)
        x = core.nhwc(x)
        n, c, h, w = x.shape
        g = min(group, n)
        cp = -(-(c + 1) // core.epc(x.dtype)) * core.epc(x.dtype)
        y = core.empty_nhwc(n, cp, h, w, x.dtype, x.device)
        stat = torch.empty(n // g, dtype=torch.float32, device=x.device)
        _native.check(_native.lib().vqk_mbstd(core.dcode(x.dtype), x.data_ptr(), 0, y.data_ptr(), stat.data_ptr(), n, h * w, c, cp, g,
                                              0, core._stream()), 'mbstd')
        ctx.save_for_backward(x)
        ctx.cfg = (g, cp)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        g, cp = ctx.cfg
        return MbstdBwdFn.apply(x, core.nhwc(dy), g, cp), None


class MbstdBwdFn(torch.autograd.Function):
    """first backward of the minibatch-stddev layer as a differentiable op (it is non-linear in x, and R1
    differentiates the backward pass)"""

    @staticmethod
    def forward(ctx, x, dy, g: int, cp: int):
        n, c, h, w = x.shape
        dx = torch.empty_like(x, memory_format=core._CL)
        _native.check(_native.lib().vqk_mbstd(core.dcode(x.dtype), x.data_ptr(), dy.data_ptr(), dx.data_ptr(), 0, n, h * w, c, cp,
                                              g, 1, core._stream()), 'mbstd_backward')
        ctx.save_for_backward(x, dy)
        ctx.cfg = (g, cp)
        return dx

    @staticmethod
    def backward(ctx, v):
        x, dy = ctx.saved_tensors
        g, cp = ctx.cfg
        n, c, h, w = x.shape
        v = core.nhwc(v)
        ddy = torch.empty_like(dy, memory_format=core._CL)
        dxx = torch.empty_like(x, memory_format=core._CL)
        _native.check(_native.lib().vqk_mbstd_double_backward(core.dcode(x.dtype), x.data_ptr(), dy.data_ptr(), v.data_ptr(),
                                                              ddy.data_ptr(), dxx.data_ptr(), n, h * w, c, cp, g, core._stream()),
                      'mbstd_double_backward')
        return dxx, ddy, None, None


class AddFn(torch.autograd.Function):
    """a + b on the HIP axpby kernel (the resnet skip add of the discriminator, discriminator.py:259)"""

    @staticmethod
    def forward(ctx, a, b):
        a, b = core.nhwc(a), core.nhwc(b)
        y = torch.empty_like(a, memory_format=core._CL)
        _native.check(_native.lib().vqk_axpby(core.dcode(a.dtype), a.data_ptr(), b.data_ptr(), y.data_ptr(), 1.0, 1.0, a.numel(),
                                              core._stream()), 'axpby')
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy




def _conv_act_cfg(x, weight, k, stride, pad, act, wgain, out_gain):
    """the ``ctx.cfg`` tuple ConvActFn.forward builds for this call (channel counts already whole 16-byte chunks)"""
    o, i = weight.shape[0], weight.shape[1]
    e = core.epc(x.dtype)
    return (k, stride, pad, act, float(wgain), float(out_gain), o, i, x.shape[1], -(-o // e) * e, x.dtype)


class DiscBlockFn(torch.autograd.Function):
    """One resnet DiscriminatorBlock (discriminator.py:233-262: y = skip(x) * sqrt(1/2) + conv1(conv0(x)) * sqrt(1/2)) as ONE
    autograd node.  Forward: the same five launches as the layer-by-layer form.  Backward, in an order autograd cannot choose:
    (1) conv1 (activation gradient, stride-2 data gradient, weight gradient), (2) the skip branch down to the block input's
    resolution, (3) conv0's activation gradient FUSED into the blur's adjoint (vqk_upfirdn2d_act_backward: blur^T(dB) is never
    stored), (4) conv0's data gradient with the skip branch's gradient added IN ITS EPILOGUE (no accumulation pass).
    First-order only: the R1 pass (autograd.grad(..., create_graph=True), loss.py:98-112) uses the layer-by-layer form."""

    @staticmethod
    def layers(x, w0, b0, w1, b1, ws, f, cfg):
        """the block layer by layer (DiscriminatorBlock.forward's un-fused form); returns the intermediates as well"""
        wg0, wg1, wgs, act, act_gain, gain, pad_blur, pad_skip = cfg
        ys = UpfirdnNhwcFn.apply(x, f, 1, 2, pad_skip, False, 1.0)
        ysk = ConvActFn.apply(ys, ws, None, 1, 1, 0, 0, wgs, gain, None)
        y0 = ConvActFn.apply(x, w0, b0, 3, 1, 1, act, wg0, act_gain, None)
        blur = UpfirdnNhwcFn.apply(y0, f, 1, 1, pad_blur, False, 1.0)
        y1 = ConvActFn.apply(blur, w1, b1, 3, 2, 0, act, wg1, act_gain * gain, None)
        return AddFn.apply(ysk, y1), ys, y0, blur, y1

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, ws, f, cfg):
        wg0, wg1, wgs, act, act_gain, gain, pad_blur, pad_skip = cfg
        x_in, x = x, core.nhwc(x)
        out, ys, y0, blur, y1 = DiscBlockFn.layers(x, w0, b0, w1, b1, ws, f, cfg)
        ctx.block_cfg = cfg
        ctx.save_for_backward(x_in, ys, y0, blur, y1, f)        # (the INPUT itself: a double backward differentiates through it)
        ctx.refs = (w0, b0, w1, b1, ws)
        ctx.cfgs = (_conv_act_cfg(x, w0, 3, 1, 1, act, wg0, act_gain), _conv_act_cfg(blur, w1, 3, 2, 0, act, wg1, act_gain * gain),
                    _conv_act_cfg(ys, ws, 1, 1, 0, 0, wgs, gain), pad_blur, pad_skip)
        return out

    @staticmethod
    def backward(ctx, g):
        x, ys, y0, blur, y1, f = ctx.saved_tensors
        w0, b0, w1, b1, ws = ctx.refs
        cfg0, cfg1, cfgs, pad_blur, pad_skip = ctx.cfgs
        need = ctx.needs_input_grad
        if torch.is_grad_enabled():
            # this backward is itself being differentiated (create_graph=True, e.g. R1 on a pass that was not announced with
            # double_backward=True): the fused passes below record nothing, so the block is evaluated again layer by layer and
            # ITS differentiable backward is used (costs one forward of the block)
            with torch.enable_grad():
                ins = [x, w0, b0, w1, b1, ws]
                out = DiscBlockFn.layers(x, w0, b0, w1, b1, ws, f, ctx.block_cfg)[0]
                idx = [i for i in range(6) if need[i] and ins[i] is not None]
                got = torch.autograd.grad(out, [ins[i] for i in idx], g, create_graph=True, allow_unused=True)
            res = [None] * 8
            for i, v in zip(idx, got):
                res[i] = v
            return tuple(res)
        x = core.nhwc(x)
        g = core.nhwc(g)
        n, c, h, w = x.shape
        fh, fw = f.shape
        # (1) conv1
        d_blur, dw1, db1 = _conv_act_backward(blur, y1, (w1, b1), cfg1, (True, need[3], need[4]), g)
        # (2) skip: 1x1 data gradient at half resolution, then the adjoint of blur + decimate (upfirdn2d.py:259-268)
        g_lo, dws, _ = _conv_act_backward(ys, None, (ws, None), cfgs, (need[0], need[5], False), g)
        u = None
        if need[0]:
            _, _, oh, ow = ys.shape
            ps = (fw - pad_skip[0] - 1, w - ow * 2 + pad_skip[0], fh - pad_skip[2] - 1, h - oh * 2 + pad_skip[2])
            u = UpfirdnNhwcFn.apply(g_lo, f, 2, 1, ps, True, 1.0)
        # (3) + (4) conv0
        _, _, bh, bw = blur.shape
        pb = (fw - pad_blur[0] - 1, w - bw + pad_blur[0], fh - pad_blur[2] - 1, h - bh + pad_blur[2])
        act = cfg0[3]

        def make_t(scale, dbsum, db_scale=1.0):
            t0 = torch.empty_like(y0, memory_format=core._CL)
            st = _native.lib().vqk_upfirdn2d_act_backward(core.dcode(x.dtype), d_blur.data_ptr(), f.data_ptr(), y0.data_ptr(), t0.data_ptr(),
                                                          n, bh, bw, c, pb[0], pb[1], pb[2], pb[3], 1, float(scale), act, h, w,
                                                          core._stream()) if (act in (2, 3) and fh == 4 and fw == 4) else -1
            if st not in (-1, _native.ERR_SHAPE):
                _native.check(st, 'upfirdn2d_act_backward')
            else:                                        # shapes outside the fused kernel: blur^T, then the activation gradient
                t0 = ActBwdFn.apply(UpfirdnNhwcFn.apply(d_blur, f, 1, 1, pb, True, 1.0), y0, act, float(scale), None)
            if dbsum is not None:
                core.raw_colsum(n * h * w, c, t0, out=dbsum, scale=db_scale)
            return t0

        gx, dw0, db0 = _conv_act_backward(x, y0, (w0, b0), cfg0, (need[0], need[1], need[2]), None, make_t=make_t, dx_residual=u)
        return gx, dw0, db0, dw1, db1, dws, None, None


class ReconLossFn(torch.autograd.Function):
    """(l1, l2) = (mean |t - r|, mean (t - r)^2) over the un-padded element count (loss.py:58-63,118-119)"""

    @staticmethod
    def forward(ctx, recon, target, denom: float):
        core._require_gpu(recon)
        recon = core.nhwc(recon)
        target = core.nhwc(target.to(torch.float32))
        sums = torch.zeros(2, dtype=torch.float32, device=recon.device)
        lib, st = _native.lib(), core._stream()
        _native.check(lib.vqk_l1_sum(core.dcode(recon.dtype), recon.data_ptr(), target.data_ptr(), recon.numel(), sums[0:1].data_ptr(), st), 'l1')
        _native.check(lib.vqk_sse(core.dcode(recon.dtype), recon.data_ptr(), target.data_ptr(), recon.numel(), sums[1:2].data_ptr(), st), 'sse')
        ctx.save_for_backward(recon, target)
        ctx.denom = denom
        return sums[0] / denom, sums[1] / denom

    @staticmethod
    def backward(ctx, d1, d2):
        recon, target = ctx.saved_tensors
        d = torch.empty_like(recon, memory_format=core._CL)
        lib, st = _native.lib(), core._stream()
        one = torch.ones((), dtype=torch.float32, device=recon.device)
        g1 = (d1 if d1 is not None else one * 0).to(torch.float32).contiguous()
        g2 = (d2 if d2 is not None else one * 0).to(torch.float32).contiguous()
        _native.check(lib.vqk_l1l2_backward(core.dcode(recon.dtype), recon.data_ptr(), target.data_ptr(), recon.numel(),
                                            1.0 / ctx.denom, 0.0, g1.data_ptr(), d.data_ptr(), 0, st), 'l1_backward')
        _native.check(lib.vqk_l1l2_backward(core.dcode(recon.dtype), recon.data_ptr(), target.data_ptr(), recon.numel(),
                                            0.0, 1.0 / ctx.denom, g2.data_ptr(), d.data_ptr(), 1, st), 'l2_backward')
        return d, None, None


class GanLossFn(torch.autograd.Function):
    """generator_loss / discriminator_loss of loss.py:11-51 on [B,1] logits"""

    @staticmethod
    def forward(ctx, logits_real, logits_fake, mode: int, which: int):
        lf = logits_fake.to(torch.float32).contiguous()
        lr = logits_real.to(torch.float32).contiguous() if logits_real is not None else None
        core._require_gpu(lf)
        loss = torch.zeros((), dtype=torch.float32, device=lf.device)
        _native.check(_native.lib().vqk_gan_loss(core._p(lr), lf.data_ptr(), lf.numel(), mode, which, loss.data_ptr(), 0, 0, 0,
                                                 core._stream()), 'gan_loss')
        ctx.save_for_backward(lr, lf)
        ctx.cfg = (mode, which, logits_fake.shape)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        lr, lf = ctx.saved_tensors
        mode, which, shape = ctx.cfg
        dfake = torch.empty_like(lf)
        dreal = torch.empty_like(lr) if lr is not None else None
        gs = dloss.to(torch.float32).contiguous()
        _native.check(_native.lib().vqk_gan_loss(core._p(lr), lf.data_ptr(), lf.numel(), mode, which, 0, core._p(dreal), dfake.data_ptr(),
                                                 gs.data_ptr(), core._stream()), 'gan_loss_backward')
        return (dreal.view(shape) if dreal is not None else None), dfake.view(shape), None, None


class SumSqFn(torch.autograd.Function):
    """sum(g^2) over all elements (R1 penalty, loss.py:108); backward 2 g * upstream"""

    @staticmethod
    def forward(ctx, gimg):
        gimg = gimg.contiguous()
        zero = torch.zeros(gimg.numel(), dtype=torch.float32, device=gimg.device)
        out = torch.zeros((), dtype=torch.float32, device=gimg.device)
        _native.check(_native.lib().vqk_sse(core.dcode(gimg.dtype), gimg.data_ptr(), zero.data_ptr(), gimg.numel(), out.data_ptr(),
                                            core._stream()), 'sse')
        ctx.save_for_backward(gimg)
        return out

    @staticmethod
    def backward(ctx, dout):
        (gimg,) = ctx.saved_tensors
        zero = torch.zeros(gimg.numel(), dtype=torch.float32, device=gimg.device)
        d = torch.empty_like(gimg)
        _native.check(_native.lib().vqk_mse_tanh_backward(core.dcode(gimg.dtype), gimg.data_ptr(), zero.data_ptr(), gimg.numel(), 1.0,
                                                          dout.to(torch.float32).contiguous().data_ptr(), 0, d.data_ptr(),
                                                          core._stream()), 'sumsq_backward')
        return d


__all__ = [_n for _n in dir() if not _n.startswith('__') and _n not in ('core', 'annotations')]
