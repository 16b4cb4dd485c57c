"""CPU restatement (plain PyTorch fp32/fp64 on CPU) of the reference's VQ-VAE train step.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Functional style: every
network is a function of a flat ``{state_dict key: tensor}`` mapping that uses the
reference's own key names, so golden vectors captured from the reference modules
can be fed in unchanged.  Each function cites the reference lines it restates
(paths relative to the reference checkout, ``vqvae/...``).

Parity pin: ``tests/test_oracle_golden.py`` checks every function here against
``tests/golden/*.npz`` (captured by ``tests/golden/make_golden.py`` from the real
reference modules imported on CPU).
"""
from __future__ import annotations

import math
from typing import Dict, Sequence

import torch
import torch.nn.functional as F

P = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------------------
# autoencoder pieces  (vqvae/modules/autoencoder.py)
# --------------------------------------------------------------------------------------
def group_norm(x, weight, bias, groups: int = 32, eps: float = 1e-6):
    """autoencoder.py:25-39 -- per (sample, group) mean, UNBIASED variance, affine (1,C,1,1)."""
    b, c, h, w = x.shape
    xg = x.reshape(b, groups, (c // groups) * h * w)
    mu = xg.mean(dim=2, keepdim=True)
    var = xg.var(dim=2, keepdim=True, unbiased=True)
    xn = ((xg - mu) / torch.sqrt(var + eps)).reshape(b, c, h, w)
    return xn * weight.reshape(1, c, 1, 1) + bias.reshape(1, c, 1, 1)


def res_block(x, p: P, pre: str):
    """autoencoder.py:63-77 -- GN,SiLU,3x3 ; GN,SiLU,3x3 ; optional 1x1 shortcut ; add."""
    r = F.silu(group_norm(x, p[pre + 'norm1.weight'], p[pre + 'norm1.bias']))
    r = F.conv2d(r, p[pre + 'conv1.weight'], None, padding=1)
    r = F.silu(group_norm(r, p[pre + 'norm2.weight'], p[pre + 'norm2.bias']))
    r = F.conv2d(r, p[pre + 'conv2.weight'], None, padding=1)
    if pre + 'conv_shortcut.weight' in p:
        x = F.conv2d(x, p[pre + 'conv_shortcut.weight'], None)
    return x + r


def downsample(x):
    """autoencoder.py:89-91"""
    return F.avg_pool2d(x, 2, 2, 0)


def upsample(x, weight, bias):
    """autoencoder.py:104-106 -- nearest-exact x2 (== pixel replication) then 3x3 conv with bias."""
    x = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    return F.conv2d(x, weight, bias, padding=1)


def encoder_forward(x, p: P, num_res_blocks: int, n_levels: int, pre: str = ''):
    """autoencoder.py:135-143 (layout of ``blocks``: per level R ResBlocks then one Downsample)."""
    x = F.conv2d(x, p[pre + 'conv_in.weight'], None, padding=1)
    i = 0
    for _ in range(n_levels):
        for _ in range(num_res_blocks):
            x = res_block(x, p, f'{pre}blocks.{i}.')
            i += 1
        x = downsample(x)
        i += 1
    for j in range(num_res_blocks):
        x = res_block(x, p, f'{pre}final_residual.{j}.')
    x = F.silu(group_norm(x, p[pre + 'norm.weight'], p[pre + 'norm.bias']))
    return F.conv2d(x, p[pre + 'conv_out.weight'], p[pre + 'conv_out.bias'])


def decoder_forward(z, p: P, num_res_blocks: int, n_levels: int, pre: str = ''):
    """autoencoder.py:172-180 (``blocks``: per level R ResBlocks then one Upsample)."""
    x = F.conv2d(z, p[pre + 'conv_in.weight'], p[pre + 'conv_in.bias'], padding=1)
    for j in range(num_res_blocks):
        x = res_block(x, p, f'{pre}initial_residual.{j}.')
    i = 0
    for _ in range(n_levels):
        for _ in range(num_res_blocks):
            x = res_block(x, p, f'{pre}blocks.{i}.')
            i += 1
        x = upsample(x, p[f'{pre}blocks.{i}.conv.weight'], p[f'{pre}blocks.{i}.conv.bias'])
        i += 1
    x = F.silu(group_norm(x, p[pre + 'norm.weight'], p[pre + 'norm.bias']))
    x = F.conv2d(x, p[pre + 'conv_out.weight'], p[pre + 'conv_out.bias'], padding=1)
    return torch.tanh(x)


# --------------------------------------------------------------------------------------
# quantizers  (vqvae/modules/vector_quantizers.py, abstract_modules/base_quantizer.py)
# --------------------------------------------------------------------------------------
def _flat(z):
    b, c, h, w = z.shape
    return z.permute(0, 2, 3, 1).reshape(b * h * w, c)


def _unflat(q, shape):
    b, c, h, w = shape
    return q.reshape(b, h, w, c).permute(0, 3, 1, 2)


def distances_std(flat_z, codebook):
    """vector_quantizers.py:37-39 -- (|z|^2 + |e|^2) - 2 z.e^T, this association."""
    return (torch.sum(flat_z ** 2, dim=1, keepdim=True) + torch.sum(codebook ** 2, dim=1)
            - 2 * torch.matmul(flat_z, codebook.t()))


def distances_entropy(flat_z, codebook):
    """vector_quantizers.py:337-340 -- (|z|^2 - 2 z.e^T) + |e|^2."""
    a2 = torch.sum(flat_z ** 2, dim=1, keepdim=True)
    b2 = torch.sum(codebook.t() ** 2, dim=0, keepdim=True)
    return a2 - 2 * torch.matmul(flat_z, codebook.t()) + b2


def vq_standard(z, codebook, beta: float):
    """vector_quantizers.py:23-61 -> (q with STE grad, idx (B,HW) int64, q_loss + e_loss)."""
    fz = _flat(z)
    idx = torch.argmin(distances_std(fz, codebook), dim=1)
    q = codebook[idx]
    e_loss = beta * F.mse_loss(q.detach(), fz)
    q_loss = F.mse_loss(q, fz.detach())
    q_ste = fz + (q - fz).detach()
    return _unflat(q_ste, z.shape), idx.reshape(z.shape[0], -1), q_loss + e_loss


def vq_ema(z, codebook, ema_count, ema_weight, beta: float, decay: float, eps: float,
           training: bool = True, batch_for_smoothing: int | None = None):
    """vector_quantizers.py:128-180.  Returns (q_ste, idx, loss, new_count, new_weight, new_codebook).

    q is taken from the PRE-update codebook (:154); count smoothing uses the image batch
    size ``b`` (:164).  ``batch_for_smoothing`` overrides b (the multi-GPU definition in
    SURVEY 8(e): global batch)."""
    b = z.shape[0] if batch_for_smoothing is None else batch_for_smoothing
    k = codebook.shape[0]
    fz = _flat(z)
    idx = torch.argmin(distances_std(fz, codebook), dim=1)
    q = codebook[idx]
    new_count, new_weight, new_cb = ema_count, ema_weight, codebook
    if training:
        with torch.no_grad():
            n_k = torch.bincount(idx, minlength=k).to(fz.dtype)
            cnt = ema_count * decay + (1 - decay) * n_k
            new_count = (cnt + eps) / (b + k * eps) * b
            dw = torch.zeros_like(codebook).index_add_(0, idx, fz.detach())
            new_weight = ema_weight * decay + (1 - decay) * dw
            new_cb = new_weight / new_count.unsqueeze(1)
    loss = beta * F.mse_loss(q.detach(), fz)
    q_ste = fz + (q - fz).detach()
    return _unflat(q_ste, z.shape), idx.reshape(z.shape[0], -1), loss, new_count, new_weight, new_cb


def entropy_term(affinity, temperature: float, loss_type: str = 'softmax'):
    """vector_quantizers.py:296-328"""
    aff = affinity / temperature
    probs = F.softmax(aff, dim=-1)
    if loss_type == 'softmax':
        target = probs
    elif loss_type == 'argmax':
        codes = torch.argmax(aff, dim=-1)
        one_hot = F.one_hot(codes, aff.shape[-1]).to(codes)
        target = probs - (probs - one_hot).detach()
    else:
        raise ValueError(loss_type)
    avg = target.mean(dim=0)
    avg_entropy = -torch.sum(avg * torch.log(avg + 1e-5))
    logp = F.log_softmax(aff + 1e-5, dim=-1)
    sample_entropy = torch.mean(-torch.sum(target * logp, dim=-1))
    return sample_entropy - avg_entropy


def vq_entropy(z, codebook, beta: float, ratio: float, temperature: float, loss_type: str = 'softmax'):
    """vector_quantizers.py:290-356"""
    fz = _flat(z)
    d = distances_entropy(fz, codebook)
    idx = torch.argmin(d, dim=1)
    q = _unflat(codebook[idx], z.shape)
    e_l = torch.mean((q.detach() - z) ** 2) * beta
    q_l = torch.mean((q - z.detach()) ** 2)
    ent = entropy_term(-d, temperature, loss_type) * ratio
    q_ste = z + (q - z).detach()
    return q_ste, idx.reshape(z.shape[0], -1), e_l + q_l + ent


def vq_gumbel(x, codebook, w_logits, b_logits, tau: float, kl_cost: float, exp_noise, hard: bool = False):
    """vector_quantizers.py:223-245.  ``exp_noise`` ~ Exp(1), same shape as x: the tensor
    ``F.gumbel_softmax`` draws first (gumbels = -log(exp_noise)); injecting it makes the op a
    pure function."""
    k = codebook.shape[0]
    logits = F.conv2d(x, w_logits, b_logits)
    g = -exp_noise.log()
    y = F.softmax((logits + g) / tau, dim=1)
    if hard:
        i = y.argmax(dim=1, keepdim=True)
        y = torch.zeros_like(y).scatter_(1, i, 1.0) - y.detach() + y
    q = torch.einsum('bnhw,nd->bdhw', y, codebook)
    qy = F.softmax(logits, dim=1)
    kl = kl_cost * torch.sum(qy * torch.log(qy * k + 1e-10), dim=1).mean()
    return q, y.argmax(dim=1), kl


def codebook_usage(index_count):
    """base_quantizer.py:63-79 -> (p, perplexity, used %)"""
    p = index_count / torch.sum(index_count)
    perplexity = torch.exp(-torch.sum(p * torch.log(p + 1e-10), dim=-1)).sum().item()
    used = torch.count_nonzero(p).item() * 100 / index_count.shape[0]
    return p, perplexity, used


# --------------------------------------------------------------------------------------
# pre/post-processing (abstract_modules/base_autoencoder.py:31-61, augmentation off)
# --------------------------------------------------------------------------------------
def preprocess(images):
    return (torch.clamp(images, 0., 1.) - 0.5) / 0.5


def postprocess(y):
    return torch.clip(y * 0.5 + 0.5, 0, 1)


# --------------------------------------------------------------------------------------
# optimizer  (vqvae/model.py:372-440)
# --------------------------------------------------------------------------------------
def decay_split(names: Sequence[str]):
    """model.py:384-396 -- decay = conv weights (4-D '.weight' that is not a norm);
    no-decay = every bias, GroupNorm weight, codebook (nn.Embedding) weight."""
    decay, no_decay = [], []
    for n in names:
        leaf = n.rsplit('.', 2)
        is_norm = len(leaf) >= 2 and leaf[-2].startswith('norm')
        if n.endswith('bias') or is_norm or n.endswith('codebook.weight'):
            no_decay.append(n)
        else:
            decay.append(n)
    return decay, no_decay


def adamw_step(p, g, v, step: int, lr: float, beta1: float, beta2: float, eps: float, wd: float, m=None):
    """torch.optim.AdamW (decoupled decay) as configured at model.py:428; returns (p, m, v)."""
    m = torch.zeros_like(p) if m is None else m
    p = p * (1 - lr * wd)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


def cosine_lr(step: int, start: int, stop: int, v0: float, v1: float):
    """scheduling_utils CosineScheduler semantics as used at model.py:183-187 (PARITY UNPINNED:
    the package is not in the reference tree; half-cosine from v0 at ``start`` to v1 at ``stop``)."""
    if step <= start:
        return v0
    if step >= stop:
        return v1
    t = (step - start) / (stop - start)
    return v1 + 0.5 * (v0 - v1) * (1 + math.cos(math.pi * t))


def linear_lr(step: int, start: int, stop: int, v0: float, v1: float):
    """LinearScheduler (model.py:175-181) -- PARITY UNPINNED, same caveat."""
    if step <= start:
        return v0
    if step >= stop:
        return v1
    return v0 + (v1 - v0) * (step - start) / (stop - start)


# --------------------------------------------------------------------------------------
# the train step  (vqvae/model.py:151-161, 232-295; MSE branch :271-275)
# --------------------------------------------------------------------------------------
def split_params(p: P):
    enc = {k[len('encoder.'):]: v for k, v in p.items() if k.startswith('encoder.')}
    dec = {k[len('decoder.'):]: v for k, v in p.items() if k.startswith('decoder.')}
    return enc, dec


def vqvae_forward(images01, p: P, num_res_blocks: int, n_levels: int, qtype: str, qparams: dict,
                  buffers: dict | None = None):
    """model.py:151-161 after preprocess_batch(training=False).  ``p`` uses LightningModule keys
    (``encoder.*``, ``decoder.*``, ``quantizer.codebook.weight``)."""
    x = preprocess(images01)
    enc, dec = split_params(p)
    z = encoder_forward(x, enc, num_res_blocks, n_levels)
    cb = p['quantizer.codebook.weight']
    extra = {}
    if qtype == 'standard':
        q, idx, ql = vq_standard(z, cb, qparams['commitment_cost'])
    elif qtype == 'ema':
        q, idx, ql, c, w, ncb = vq_ema(z, cb, buffers['ema_count'], buffers['ema_weight'],
                                       qparams['commitment_cost'], qparams['decay'], qparams['epsilon'],
                                       batch_for_smoothing=qparams.get('global_batch'))
        extra = dict(ema_count=c, ema_weight=w, codebook=ncb)
    elif qtype == 'entropy':
        q, idx, ql = vq_entropy(z, cb, qparams['commitment_cost'], qparams['ent_loss_ratio'],
                                qparams['ent_temperature'], qparams.get('ent_loss_type', 'softmax'))
    else:
        raise ValueError(qtype)
    recon = decoder_forward(q, dec, num_res_blocks, n_levels)
    return x, z, recon, ql, idx, extra


def train_step_mse(images01, p: P, num_res_blocks: int, n_levels: int, qtype: str, qparams: dict,
                   buffers: dict | None = None):
    """model.py:232-295 MSE branch: ae_loss = q_loss + mse(recon, images); returns a dict with
    recon, idx, the loss terms and d(ae_loss)/d(param) for every trainable tensor."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    if qtype == 'ema':
        leaves['quantizer.codebook.weight'].requires_grad_(False)   # vector_quantizers.py:114
    x, z, recon, ql, idx, extra = vqvae_forward(images01, leaves, num_res_blocks, n_levels, qtype, qparams, buffers)
    l2 = F.mse_loss(recon, x)
    loss = ql + l2
    names = [k for k, v in leaves.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
    return dict(recon=recon.detach(), z=z.detach(), idx=idx, q_loss=ql.detach(), l2=l2.detach(),
                loss=loss.detach(), grads={k: g for k, g in zip(names, grads) if g is not None}, extra=extra)


# --------------------------------------------------------------------------------------
# StyleGAN2 custom ops, reference 'ref' twins
# --------------------------------------------------------------------------------------
def bias_act(x, b=None, dim: int = 1, act: str = 'linear', alpha: float = 0.2, gain: float = 1.0, clamp: float = -1.0):
    """stylegan2_discriminator/utils/ops/bias_act.py:94-123 (linear / lrelu only, the on-path variants)."""
    if b is not None:
        x = x + b.reshape([-1 if i == dim else 1 for i in range(x.ndim)])
    if act == 'lrelu':
        x = F.leaky_relu(x, alpha)
    elif act != 'linear':
        raise ValueError(act)
    if gain != 1:
        x = x * gain
    if clamp >= 0:
        x = x.clamp(-clamp, clamp)
    return x


def upfirdn2d(x, f, up=(1, 1), down=(1, 1), pad=(0, 0, 0, 0), flip_filter: bool = False, gain: float = 1.0):
    """stylegan2_discriminator/utils/ops/upfirdn2d.py:169-208: zero-insert upsample, pad/crop,
    true convolution with the 2-D FIR ``f`` (correlation when flip_filter), decimate."""
    n, c, h, w = x.shape
    upx, upy = up
    downx, downy = down
    px0, px1, py0, py1 = pad
    y = x.reshape(n, c, h, 1, w, 1)
    y = F.pad(y, [0, upx - 1, 0, 0, 0, upy - 1]).reshape(n, c, h * upy, w * upx)
    y = F.pad(y, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    y = y[:, :, max(-py0, 0): y.shape[2] - max(-py1, 0), max(-px0, 0): y.shape[3] - max(-px1, 0)]
    k = (f * gain).to(x.dtype)
    if not flip_filter:
        k = k.flip([0, 1])
    y = F.conv2d(y, k[None, None].repeat(c, 1, 1, 1), groups=c)
    return y[:, :, ::downy, ::downx]


def augment_crop_flip(images, box, flip):
    """Training augmentation of base_autoencoder.py:20-22,44-48 as a pure function of the draws: per-sample crop
    box = (x0, y0, w, h) resampled to the full size with corner-aligned bilinear interpolation (kornia's crop warp),
    optional horizontal flip, then clamp + Normalize(0.5, 0.5).  kornia itself is not importable here: the
    interpolation convention is restated, the random generator is unpinned."""
    n, c, h, w = images.shape
    img = torch.clamp(images, 0., 1.)
    out = torch.empty_like(img)
    for b in range(n):
        x0, y0, bw, bh = [float(v) for v in box[b]]
        xs = x0 + torch.arange(w, dtype=torch.float32) * ((bw - 1.0) / (w - 1) if w > 1 else 0.0)
        ys = y0 + torch.arange(h, dtype=torch.float32) * ((bh - 1.0) / (h - 1) if h > 1 else 0.0)
        if int(flip[b]):
            xs = xs.flip(0)
        xs, ys = xs.clamp(0, w - 1), ys.clamp(0, h - 1)
        ix = xs.floor().long(); iy = ys.floor().long()
        fx = (xs - ix).view(1, 1, w); fy = (ys - iy).view(1, h, 1)
        ix1 = (ix + 1).clamp(max=w - 1); iy1 = (iy + 1).clamp(max=h - 1)
        im = img[b]
        c00 = im[:, iy][:, :, ix]; c01 = im[:, iy][:, :, ix1]
        c10 = im[:, iy1][:, :, ix]; c11 = im[:, iy1][:, :, ix1]
        top = c00 + fx * (c01 - c00); bot = c10 + fx * (c11 - c10)
        out[b] = top + fy * (bot - top)
    return (out - 0.5) / 0.5


# --------------------------------------------------------------------------------------
# Test-loop metrics (vqvae/model.py:491-553).  The reference calls torchmetrics (not vendored, no pinned version in the
# tree: parity UNPINNED); these restate the library's published functional algorithm.
# --------------------------------------------------------------------------------------
def metric_gaussian_window(kernel_size: int = 11, sigma: float = 1.5):
    dist = torch.arange((1 - kernel_size) / 2, (1 + kernel_size) / 2, 1.0, dtype=torch.float32)
    g = torch.exp(-torch.pow(dist / sigma, 2) / 2)
    g = (g / g.sum()).unsqueeze(0)
    return torch.matmul(g.t(), g)


def metric_ssim_per_image(preds, target, sigma: float = 1.5, k1: float = 0.01, k2: float = 0.03):
    """torchmetrics.functional.image.ssim `_ssim_update` with gaussian_kernel=True, data_range=None: reflect-pad by
    (k-1)/2, depthwise Gaussian conv of (p, t, p*p, t*t, p*t), SSIM map, crop the pad again, mean per image."""
    ks = int(3.5 * sigma + 0.5) * 2 + 1
    pad = (ks - 1) // 2
    c = preds.shape[1]
    data_range = max(preds.max() - preds.min(), target.max() - target.min())
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    p = F.pad(preds, (pad, pad, pad, pad), mode='reflect')
    t = F.pad(target, (pad, pad, pad, pad), mode='reflect')
    kernel = metric_gaussian_window(ks, sigma).to(preds.dtype).expand(c, 1, ks, ks)
    outs = F.conv2d(torch.cat((p, t, p * p, t * t, p * t)), kernel, groups=c).split(preds.shape[0])
    mu_p_sq, mu_t_sq, mu_pt = outs[0] ** 2, outs[1] ** 2, outs[0] * outs[1]
    s_p, s_t, s_pt = outs[2] - mu_p_sq, outs[3] - mu_t_sq, outs[4] - mu_pt
    full = ((2 * mu_pt + c1) * (2 * s_pt + c2)) / ((mu_p_sq + mu_t_sq + c1) * (s_p + s_t + c2))
    return full[..., pad:-pad, pad:-pad].reshape(preds.shape[0], -1).mean(-1)


def metric_epoch(batches):
    """batches: iterable of (preds, target) in [0, 1] -> dict(mse, psnr, ssim) as the three torchmetrics objects of
    model.py:494-496 report them after ``update`` on every batch (PSNR: data_range = running max - min of target)."""
    sse, n, tmin, tmax, ssim_sum, imgs = 0.0, 0, float('inf'), float('-inf'), 0.0, 0
    for p, t in batches:
        p, t = p.double(), t.double()
        sse += ((p - t) ** 2).sum().item(); n += p.numel()
        tmin, tmax = min(tmin, t.min().item()), max(tmax, t.max().item())
        ssim_sum += metric_ssim_per_image(p, t).sum().item(); imgs += p.shape[0]
    mse = sse / n
    import math
    return dict(mse=mse, psnr=10.0 * math.log10((tmax - tmin) ** 2 / mse), ssim=ssim_sum / imgs)


# --------------------------------------------------------------------------------------
# Entropy quantizer at sizes where [N, K] does not fit comfortably: the same arithmetic as ``vq_entropy``
# (vector_quantizers.py:290-356), evaluated in row chunks with the closed-form backward of SURVEY Appendix B.
# Pinned against the reference's own output at K=8192, N=4096 (tests/golden/full_entropy.npz).
# --------------------------------------------------------------------------------------
def vq_entropy_chunked(z, codebook, beta: float, ratio: float, temperature: float, dq=None, chunk: int = 2048):
    """-> dict(idx, loss, dz, de): loss = (1+beta) mse(q, z) + ratio (mean_i H(p_i) - H(mean_i p_i)), p = softmax(-d/T),
    d = (|z|^2 - 2 z.e) + |e|^2 (:337-340).  dz / de are d(loss)/dz (+ the straight-through ``dq``) and d(loss)/dE.
    Two sweeps over row chunks: (A) argmin, p-bar, sum_i H(p_i); (B) the cotangent of the distances."""
    fz = _flat(z).contiguous()
    n, d_ = fz.shape
    k = codebook.shape[0]
    e2 = torch.sum(codebook.t() ** 2, dim=0, keepdim=True)
    idx = torch.empty(n, dtype=torch.int64)
    pbar = torch.zeros(k, dtype=torch.float64)
    hsum = 0.0
    for s in range(0, n, chunk):
        c = fz[s:s + chunk]
        dist = torch.sum(c ** 2, dim=1, keepdim=True) - 2 * torch.matmul(c, codebook.t()) + e2
        idx[s:s + chunk] = torch.argmin(dist, dim=1)
        a = -dist / temperature
        logp = F.log_softmax(a, dim=-1)
        p = logp.exp()
        pbar += p.double().sum(0)
        hsum += -(p * logp).double().sum().item()
    pbar = (pbar / n).float()
    q = codebook[idx]
    mse = torch.mean((q - fz) ** 2)
    avg_entropy = -torch.sum(pbar * torch.log(pbar + 1e-5))
    loss = (1.0 + beta) * mse + ratio * (hsum / n - avg_entropy)
    # backward: d loss / d a_ik = ratio/N * p_ik * (-(l_ik + h_i) - (g_k - gbar_i)),  g_k = -(log(pbar+1e-5) + pbar/(pbar+1e-5))
    g = -(torch.log(pbar + 1e-5) + pbar / (pbar + 1e-5))
    dz = torch.zeros_like(fz)
    de = torch.zeros_like(codebook)
    nd = float(n * d_)
    for s in range(0, n, chunk):
        c = fz[s:s + chunk]
        dist = torch.sum(c ** 2, dim=1, keepdim=True) - 2 * torch.matmul(c, codebook.t()) + e2
        logp = F.log_softmax(-dist / temperature, dim=-1)
        p = logp.exp()
        h = -(p * logp).sum(1, keepdim=True)
        gbar = (p * g).sum(1, keepdim=True)
        da = (ratio / n) * p * (-(logp + h) - (g - gbar))
        dd = -da / temperature
        # d = |z|^2 - 2 z.e + |e|^2:  dz += 2 z rowsum(dd) - 2 dd @ E ; dE += -2 dd^T @ z + 2 E colsum(dd)
        dz[s:s + chunk] += 2 * c * dd.sum(1, keepdim=True) - 2 * dd @ codebook
        de += -2 * dd.t() @ c + 2 * codebook * dd.sum(0).unsqueeze(1)
    qi = q
    dz += beta * 2 * (fz - qi) / nd
    de.index_add_(0, idx, 2 * (qi - fz) / nd)
    dz = _unflat(dz, z.shape)
    if dq is not None:
        dz = dz + dq
    return dict(idx=idx.reshape(z.shape[0], -1), loss=loss, dz=dz, de=de)
