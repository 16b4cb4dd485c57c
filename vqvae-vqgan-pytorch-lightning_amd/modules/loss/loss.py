"""VQ-GAN criteria on the vqk kernels; same names / signatures / return tuples as the reference's
``vqvae/modules/loss/loss.py`` (generator_loss :11-26, discriminator_loss :29-51, VQLPIPSWithDiscriminator :54-164).

R1 regularisation (loss.py:98-112) is built on double-differentiable backward pieces (``ops.ConvDgradFn``,
``ops.ActBwdFn``, ``ops.MbstdBwdFn``, ``ops.UpfirdnNhwcFn``).  ``VQLPIPS`` (AlexNet ablation) is out of scope."""
import os

import torch
from torch import nn

from ... import ops
from .discriminator import Discriminator
from .lpips import LPIPS
from ..autoencoder import _to_internal

BATCHED_DISC = os.environ.get('VQK_BATCHED_DISC', '1') == '1'    # discriminator step: real | fake in one pass (0: two passes)
# The reference evaluates D(fake) twice per step on the SAME reconstruction with the SAME discriminator weights: for the
# generator loss (loss.py:63 of the reference) and again, detached, for the discriminator loss (:83) -- the discriminator's
# optimizer steps only after both.  1: the second evaluation is not run; the discriminator loss backpropagates through the
# activations the first one saved (same logits bit for bit, same gradients up to the summation order of the weight gradients:
# two batches of n instead of one of 2n).  0: the two-pass form.
SHARE_FAKE_PASS = os.environ.get('VQK_SHARE_FAKE_PASS', '1') == '1'
LPIPS_SIDE_STREAM = os.environ.get('VQK_LPIPS_SIDE_STREAM', '1') == '1'
DISC_REAL_SIDE_STREAM = os.environ.get('VQK_DISC_REAL_SIDE_STREAM', '1') == '1'

_MODE = {'hinge': 0, 'non-saturating': 1}


def generator_loss(logits: torch.Tensor, loss_type: str = 'hinge'):
    if loss_type not in _MODE:
        raise ValueError(f'unknown loss_type: {loss_type}')
    return ops.GanLossFn.apply(None, logits, _MODE[loss_type], 0)


def discriminator_loss(logits_real: torch.Tensor, logits_fake: torch.Tensor, loss_type: str = 'hinge'):
    if loss_type not in _MODE:
        raise ValueError(f'unknown loss_type: {loss_type}')
    return ops.GanLossFn.apply(logits_real, logits_fake, _MODE[loss_type], 1)


class VQLPIPSWithDiscriminator(nn.Module):
    shared_fake_logits = None                # (class defaults: a criterion assembled without __init__ has them too)
    _shared_for = None
    used_aux_streams = ()

    def __init__(self, image_size: int, l1_weight: float, l2_weight: float, perc_weight: float, adversarial_conf: dict):
        super().__init__()
        self.l1_weight, self.l2_weight, self.perceptual_weight = l1_weight, l2_weight, perc_weight
        self.perceptual_loss = LPIPS(net_type='vgg')
        self.discriminator = Discriminator(image_size)
        self.adversarial_start_epoch = adversarial_conf['start_epoch']
        self.adversarial_loss_type = adversarial_conf['loss_type']
        self.generator_weight = adversarial_conf['g_weight']
        self.use_adaptive_g_weight = adversarial_conf['use_adaptive']
        self.r1_regularization_cost = adversarial_conf['r1_reg_weight']
        self.r1_regularization_every = adversarial_conf['r1_reg_every']
        self.shared_fake_logits = None       # D(fake) of the last generator half, kept for the discriminator half (SHARE_FAKE_PASS)
        self._shared_for = None              # ... and the reconstruction tensor (weak reference) it was computed on
        self.used_aux_streams = []           # side streams the last forward_* issued work on: the caller joins them after ITS backward

    def _share(self, logits, reconstructions):
        import weakref
        self.shared_fake_logits = logits
        self._shared_for = None if logits is None else weakref.ref(reconstructions)

    def join_aux_streams(self) -> None:
        """the current stream waits for every side stream the last forward used.  Autograd runs a node's backward on the stream
        of its forward, and kernels that add weight / bias gradients straight into the optimizer's arena return None to autograd:
        no AccumulateGrad node, so the engine's end-of-backward stream sync does not cover them -- the optimizer (current stream)
        must not read the arena while a side stream's atomics are in flight.  One event wait per stream; legal under capture."""
        cur = torch.cuda.current_stream()
        for st in self.used_aux_streams:
            cur.wait_stream(st)
        self.used_aux_streams = []

    def calculate_adaptive_weight(self, nll_loss, g_loss, last_layer):
        """lambda = clamp(|d nll / d W_last| / (|d g / d W_last| + 1e-8), 0, 1e4) * g_weight   (loss.py:80-96)"""
        with ops.no_direct_grad(), ops.no_param_grads():     # (only d/d last_layer is asked for: no discriminator weight gradients)
            nll_grads = ops.autograd_grad(nll_loss, last_layer, retain_graph=True)[0].detach()
            g_grads = ops.autograd_grad(g_loss, last_layer, retain_graph=True)[0].detach()
        w = torch.norm(nll_grads, p=2) / (torch.norm(g_grads, p=2) + 1e-8)
        return torch.clamp(w, 0.0, 1e4).detach() * self.generator_weight

    def forward_autoencoder(self, quantizer_loss, images, reconstructions, current_epoch: int, last_layer=None):
        n, c, h, w = reconstructions.shape
        l1_loss, l2_loss = ops.ReconLossFn.apply(reconstructions, images, float(n * 3 * h * w))
        adversarial = current_epoch >= self.adversarial_start_epoch
        side = None
        self.used_aux_streams = []
        if LPIPS_SIDE_STREAM and adversarial and reconstructions.is_cuda:
            # LPIPS (two VGG16 passes) and the discriminator pass over the reconstruction are independent until their gradients
            # meet at the reconstruction: LPIPS is issued on a second stream -- autograd runs its backward on that stream too --
            # so the under-filled small-map launches and the bandwidth-bound passes of one chain run next to the other's convs
            main, side = torch.cuda.current_stream(), ops.aux_stream(reconstructions.device, 'lpips')
            side.wait_stream(main)
            self.used_aux_streams.append(side)
            with torch.cuda.stream(side):
                p_loss = self.perceptual_loss(images, reconstructions)
        else:
            p_loss = self.perceptual_loss(images, reconstructions)
        if adversarial:
            logits_fake = self.discriminator(reconstructions)
        if side is not None:
            main.wait_stream(side)
        nll_loss = l1_loss * self.l1_weight + l2_loss * self.l2_weight + p_loss * self.perceptual_weight
        if adversarial:
            self._share(logits_fake if (SHARE_FAKE_PASS and self.training) else None, reconstructions)
            g_loss = generator_loss(logits_fake, loss_type=self.adversarial_loss_type)
            if self.training and self.use_adaptive_g_weight:
                g_weight = self.calculate_adaptive_weight(p_loss, g_loss, last_layer=last_layer)   # p_loss, as the reference
            else:
                g_weight = self.generator_weight
            loss = nll_loss + g_loss * g_weight + quantizer_loss
        else:
            self._share(None, None)                          # (no discriminator pass in this step: nothing to share, nothing kept alive)
            g_loss = torch.zeros_like(nll_loss, requires_grad=False)
            g_weight = 0.
            loss = nll_loss + quantizer_loss
        return loss, l1_loss, l2_loss, p_loss, g_loss, g_weight

    def calculate_r1_regularization_term(self, logits_real, images, compute_r1: bool):
        """w * mean_b sum_chw (d sum(logits_real) / d images)^2, differentiable w.r.t. the discriminator weights.
        Weight gradients of the inner autograd.grad are simply not requested (the reference wraps it in
        ``no_weight_gradients``)."""
        if not compute_r1:
            return 0.
        gradients, = torch.autograd.grad(outputs=logits_real.sum(), inputs=images, create_graph=True)
        return self.r1_regularization_cost * ops.SumSqFn.apply(gradients) / gradients.shape[0]

    def forward_discriminator(self, images, reconstructions, current_epoch: int, current_step: int):
        if current_epoch >= self.adversarial_start_epoch:
            compute_r1 = (self.training and current_step % self.r1_regularization_every == 0
                          and self.r1_regularization_cost is not None)
            images = images.detach().requires_grad_(compute_r1)
            self.used_aux_streams = []
            # the logits of the generator half are reused only for the VERY tensor they were computed on (object identity): any
            # other reconstruction -- a caller following the reference loop with its own tensors -- takes the two-pass form, whose
            # loss supports a plain loss.backward()
            shared = self.shared_fake_logits if SHARE_FAKE_PASS else None
            if shared is not None and not (self._shared_for is not None and self._shared_for() is reconstructions):
                shared = None
            if shared is not None:
                # (the caller restricts the backward to the discriminator's parameters: the graph behind `shared` also leads
                # into the decoder -- model.VQVAE._gan_disc_half)
                logits_fake = shared
                # Not in R1 steps: there the real pass is layer-by-layer and differentiated twice, so one parameter can receive a
                # RETURNED gradient (autograd's read-modify-write accumulation) from one chain while the other chain's kernels add
                # to it atomically from the other stream -- lost updates (seen as a 10-80 % error of b8.conv0.weight's gradient in
                # 8 of 30 golden steps).  Without R1 both chains run the same nodes: a parameter is accumulated atomically by both
                # or returned by both (one AccumulateGrad node, one stream).  Not in deterministic mode either (fixed summation order).
                if DISC_REAL_SIDE_STREAM and images.is_cuda and not ops.DETERMINISTIC and not compute_r1:
                    # the real pass on a second stream: autograd runs its backward there, next to the backward of the fake pass
                    # (whose forward ran on the main stream in the generator half) -- two independent chains through the same weights
                    main, side = torch.cuda.current_stream(), ops.aux_stream(images.device, 'disc_real')
                    side.wait_stream(main)
                    self.used_aux_streams.append(side)
                    with torch.cuda.stream(side):
                        logits_real = self.discriminator(images, double_backward=compute_r1)
                        r1_term = self.calculate_r1_regularization_term(logits_real, images, compute_r1)
                    main.wait_stream(side)
                    d_loss = discriminator_loss(logits_real, logits_fake, loss_type=self.adversarial_loss_type)
                    return d_loss + r1_term, d_loss, r1_term
                logits_real = self.discriminator(images, double_backward=compute_r1)
            elif BATCHED_DISC and not compute_r1 and images.shape[0] == reconstructions.shape[0] and images.shape[2:] == reconstructions.shape[2:]:
                # real | fake through the discriminator as ONE batch (same logits: only the minibatch-stddev layer couples
                # samples and it groups inside each half): half the launches, no per-parameter gradient accumulation
                dt = self.discriminator.compute_dtype
                both = torch.cat([_to_internal(images, dt), _to_internal(reconstructions.detach(), dt)], 0)
                logits_real, logits_fake = self.discriminator(both, halves=2).chunk(2, 0)
            else:
                logits_real = self.discriminator(images, double_backward=compute_r1)
                logits_fake = self.discriminator(reconstructions.detach())
            d_loss = discriminator_loss(logits_real, logits_fake, loss_type=self.adversarial_loss_type)
            r1_term = self.calculate_r1_regularization_term(logits_real, images, compute_r1)
            return d_loss + r1_term, d_loss, r1_term
        return None, torch.zeros((1,), device=images.device), 0.
