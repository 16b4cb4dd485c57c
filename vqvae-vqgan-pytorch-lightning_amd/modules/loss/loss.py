"""VQ-GAN criteria on the vqk kernels; same names / signatures / return tuples as the reference's
``vqvae/modules/loss/loss.py`` (generator_loss :11-26, discriminator_loss :29-51, VQLPIPSWithDiscriminator :54-164).

Not built yet: R1 regularisation (needs the double backward of the conv / upfirdn / lrelu chain -- SURVEY "hard
parts"; ``r1_reg_weight`` must be None) and the adaptive generator weight (two extra autograd.grad passes to the last
decoder layer); ``VQLPIPS`` (AlexNet ablation) is out of scope."""
import torch
from torch import nn

from ... import ops
from .discriminator import Discriminator
from .lpips import LPIPS

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
        if self.use_adaptive_g_weight:
            raise NotImplementedError('adaptive generator weight (loss.py:80-96) is not built yet')
        if self.r1_regularization_cost is not None:
            raise NotImplementedError('R1 regularisation (loss.py:98-112) needs double backward: not built yet; '
                                      'set adversarial_params.r1_reg_weight to null')

    def forward_autoencoder(self, quantizer_loss, images, reconstructions, current_epoch: int, last_layer=None):
        n, c, h, w = reconstructions.shape
        l1_loss, l2_loss = ops.ReconLossFn.apply(reconstructions, images, float(n * 3 * h * w))
        p_loss = self.perceptual_loss(images, reconstructions)
        nll_loss = l1_loss * self.l1_weight + l2_loss * self.l2_weight + p_loss * self.perceptual_weight
        if current_epoch >= self.adversarial_start_epoch:
            logits_fake = self.discriminator(reconstructions)
            g_loss = generator_loss(logits_fake, loss_type=self.adversarial_loss_type)
            g_weight = self.generator_weight
            loss = nll_loss + g_loss * g_weight + quantizer_loss
        else:
            g_loss = torch.zeros_like(nll_loss, requires_grad=False)
            g_weight = 0.
            loss = nll_loss + quantizer_loss
        return loss, l1_loss, l2_loss, p_loss, g_loss, g_weight

    def forward_discriminator(self, images, reconstructions, current_epoch: int, current_step: int):
        if current_epoch >= self.adversarial_start_epoch:
            logits_real = self.discriminator(images)
            logits_fake = self.discriminator(reconstructions.detach())
            d_loss = discriminator_loss(logits_real, logits_fake, loss_type=self.adversarial_loss_type)
            return d_loss, d_loss, 0.
        return None, torch.zeros((1,), device=images.device), 0.
