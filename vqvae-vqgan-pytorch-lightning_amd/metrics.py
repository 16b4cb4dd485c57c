"""Reconstruction metrics of the reference's test loop (vqvae/model.py:491-553), device-resident: MSE, PSNR and SSIM as
torchmetrics defines them (MeanSquaredError, PeakSignalNoiseRatio(data_range=None), StructuralSimilarityIndexMeasure()
with its defaults: Gaussian 11x11 window, sigma 1.5, k1 = 0.01, k2 = 0.03, data_range from the batch).  torchmetrics is
not part of the reference tree, so its published algorithm is restated (oracle/vqvae_oracle.py::metric_*) -- parity
unpinned.  rFID needs the pretrained Inception network (no weights offline) and is not computed.

State lives on the device and ``update`` issues two HIP kernels (``vqk_pair_stats``, ``vqk_ssim_sum``) with no host
synchronisation; ``compute`` reads the totals once."""
from __future__ import annotations

import math

import torch

from . import _native


def gaussian_window(kernel_size: int = 11, sigma: float = 1.5) -> torch.Tensor:
    """normalised 2-D Gaussian as the outer product of the 1-D kernel (torchmetrics ``_gaussian_kernel_2d``)"""
    dist = torch.arange((1 - kernel_size) / 2, (1 + kernel_size) / 2, 1.0, dtype=torch.float32)
    g = torch.exp(-torch.pow(dist / sigma, 2) / 2)
    g = (g / g.sum()).unsqueeze(0)
    return torch.matmul(g.t(), g)


class ReconstructionMetrics:
    def __init__(self, device, sigma: float = 1.5, k1: float = 0.01, k2: float = 0.03):
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('vqk: the test-loop metrics run on the GPU only (HIP kernels, no CPU fallback)')
        self.ksize = int(3.5 * sigma + 0.5) * 2 + 1                 # torchmetrics: 11 for sigma = 1.5
        self.window = gaussian_window(self.ksize, sigma).to(self.device).contiguous()
        self.k1, self.k2 = float(k1), float(k2)
        inf = float('inf')
        self.total = torch.tensor([0.0, inf, -inf, inf, -inf], dtype=torch.float32, device=self.device)  # sse, t/p range
        self.ssim_sum = torch.zeros((), dtype=torch.float32, device=self.device)
        self.n_elems = 0
        self.n_images = 0

    @torch.no_grad()
    def update(self, preds: torch.Tensor, target: torch.Tensor) -> None:
        """preds, target: (B, C, H, W) in [0, 1]"""
        p = preds.detach().to(torch.float32).contiguous()
        t = target.detach().to(torch.float32).contiguous()
        if p.shape != t.shape or p.dim() != 4 or not p.is_cuda:
            raise RuntimeError('vqk: metrics expect two CUDA tensors of the same (B, C, H, W) shape')
        b, c, h, w = p.shape
        lib, st = _native.lib(), torch.cuda.current_stream().cuda_stream
        inf = float('inf')
        batch = torch.tensor([0.0, inf, -inf, inf, -inf], dtype=torch.float32, device=p.device)
        _native.check(lib.vqk_pair_stats(p.data_ptr(), t.data_ptr(), p.numel(), batch.data_ptr(), st), 'pair_stats')
        per_image = torch.zeros(b, dtype=torch.float32, device=p.device)
        _native.check(lib.vqk_ssim_sum(p.data_ptr(), t.data_ptr(), b, c, h, w, self.window.data_ptr(), self.ksize,
                                       batch.data_ptr(), self.k1, self.k2, per_image.data_ptr(), st), 'ssim_sum')
        valid = c * (h - self.ksize + 1) * (w - self.ksize + 1)
        self.ssim_sum += (per_image / valid).sum()
        self.total[0] += batch[0]
        self.total[1] = torch.minimum(self.total[1], batch[1])
        self.total[2] = torch.maximum(self.total[2], batch[2])
        self.n_elems += p.numel()
        self.n_images += b

    def compute(self) -> dict:
        sse, tmin, tmax = (float(v) for v in self.total[:3].tolist())
        mse = sse / max(self.n_elems, 1)
        data_range = tmax - tmin
        psnr = 10.0 * math.log10(data_range * data_range / mse) if mse > 0 and data_range > 0 else float('inf')
        return dict(mse=mse, psnr=psnr, ssim=float(self.ssim_sum) / max(self.n_images, 1))
