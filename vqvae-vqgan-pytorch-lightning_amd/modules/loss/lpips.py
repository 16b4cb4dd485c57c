"""LPIPS v0.1 with the VGG16 backbone on the vqk kernels (reference: vqvae/modules/loss/lpips_pytorch/modules/
lpips.py:8-38, networks.py:24-97, utils.py:6-30).

``state_dict`` keys follow the reference (``net.layers.{i}.weight|bias`` with torchvision's ``vgg16.features``
indices, ``net.mean``, ``net.std``, ``lin.{k}.1.weight``), so pretrained torchvision / LPIPS weights load when they
are available.  Offline there is nothing to download: the constructor leaves the backbone at its PyTorch default
initialisation and the lin layers at ones -- the architecture and the LPIPS arithmetic are what is pinned
(PARITY UNPINNED for the pretrained values).  Everything is frozen, as in the reference."""
import torch
from torch import nn

from ... import ops
from ..autoencoder import Conv2d, _to_internal

# torchvision cfg "D": (feature index, cin, cout); 'M' = max-pool.  Taps after the ReLU of indices 3, 8, 15, 22, 29.
_VGG16 = [(0, 3, 64), (2, 64, 64), 'M', (5, 64, 128), (7, 128, 128), 'M', (10, 128, 256), (12, 256, 256), (14, 256, 256),
          'M', (17, 256, 512), (19, 512, 512), (21, 512, 512), 'M', (24, 512, 512), (26, 512, 512), (28, 512, 512)]
_TAPS_AFTER = (2, 7, 14, 21, 28)        # conv index whose ReLU output is tapped (1-indexed layers 4, 9, 16, 23, 30)


class VGG16(nn.Module):
    n_channels_list = [64, 128, 256, 512, 512]

    def __init__(self):
        super().__init__()
        self.compute_dtype = torch.float32
        self.layers = nn.ModuleDict({str(e[0]): Conv2d(e[1], e[2], 3, bias=True) for e in _VGG16 if e != 'M'})
        self.register_buffer('mean', torch.Tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer('std', torch.Tensor([.458, .448, .450])[None, :, None, None])
        for p in self.parameters():
            p.requires_grad = False

    def _zscore(self, c: int, device):
        """per-channel (1 / std, -mean / std) zero-padded to the internal channel count: derived ONCE per (buffers, device) -- the
        eight fill / copy / elementwise launches that built them ran in every VGG pass of every step"""
        key = (c, device, self.mean._version, self.std._version, self.mean.data_ptr(), self.std.data_ptr())
        if getattr(self, '_zs_key', None) != key:
            scale = torch.zeros(c, dtype=torch.float32, device=device)
            shift = torch.zeros(c, dtype=torch.float32, device=device)
            scale[:3] = 1.0 / self.std.reshape(-1).to(device)
            shift[:3] = -self.mean.reshape(-1).to(device) / self.std.reshape(-1).to(device)
            self._zs_key, self._zs = key, (scale, shift)
        return self._zs

    def forward(self, x):
        """-> the 5 tap activations (un-normalised; the tap kernel does the channel normalisation)"""
        ops.set_conv_products(getattr(self, 'conv_products', 'fp32'))
        x = _to_internal(x, self.compute_dtype)
        scale, shift = self._zscore(x.shape[1], x.device)
        x = ops.ChannelAffineFn.apply(x, scale, shift)          # z-score; pad channels stay zero
        taps = []
        for e in _VGG16:
            if e == 'M':
                x = ops.MaxPool2x2Fn.apply(x)
                continue
            conv = self.layers[str(e[0])]
            x = ops.conv_act(x, conv.weight, conv.bias, k=3, act='relu')
            if e[0] in _TAPS_AFTER:
                taps.append(x)
        return taps


class LPIPS(nn.Module):
    def __init__(self, net_type: str = 'vgg', version: str = '0.1'):
        super().__init__()
        if net_type != 'vgg' or version != '0.1':
            raise NotImplementedError("only LPIPS v0.1 / 'vgg' is on the VQ-GAN path (loss.py:66)")
        self.net = VGG16()
        self.lin = nn.ModuleList([nn.Sequential(nn.Identity(), nn.Conv2d(nc, 1, 1, 1, 0, bias=False))
                                  for nc in self.net.n_channels_list])
        for p in self.lin.parameters():
            nn.init.ones_(p)
            p.requires_grad = False

    def forward(self, x: torch.Tensor, y: torch.Tensor):
        """x: target images (no gradient), y: reconstructions"""
        with torch.no_grad():
            fx = self.net(x)
        fy = self.net(y)
        ws = [lin[1].weight.detach().reshape(-1).to(torch.float32).contiguous() for lin in self.lin]
        total = ops.LpipsTapsFn.apply(len(ws), *fx, *fy, *ws)   # [B]: the five taps' terms, one accumulation buffer
        return total.mean()
