#!/usr/bin/env python
"""The entropy quantizer's distance matrix at config-5 size (N = 16,384, K = 8,192, D = 256) on both kernels (tuning slot VQ_LDS
0 / 1), three launches each -- the target of tools/pmc_entropy_dist.sh"""
import importlib, os, sys
import torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')
ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
lib = native.lib()
n, k, d = 16384, 8192, 256
g = torch.Generator(device='cuda').manual_seed(0)
z = torch.randn(n, d, device='cuda', generator=g)
e = torch.randn(k, d, device='cuda', generator=g)
f32 = dict(dtype=torch.float32, device='cuda')
z2, e2 = torch.empty(n, **f32), torch.empty(k, **f32)
st = ops._stream()
lib.vqk_row_sqnorm_f32(z.data_ptr(), n, d, z2.data_ptr(), st)
lib.vqk_row_sqnorm_f32(e.data_ptr(), k, d, e2.data_ptr(), st)
idx = torch.empty(n, dtype=torch.int64, device='cuda')
dm = torch.empty(n, k, **f32)
lse, hrow, hsum = torch.empty(n, **f32), torch.empty(n, **f32), torch.zeros(1, **f32)
for slot in (0, 1):
    lib.vqk_set_tuning(b'VQ_LDS', slot)
    for _ in range(3):
        native.check(lib.vqk_vq_distances_stats_f32(z.data_ptr(), e.data_ptr(), z2.data_ptr(), e2.data_ptr(), n, k, d, 1, idx.data_ptr(),
                                                    dm.data_ptr(), 0.05, lse.data_ptr(), hrow.data_ptr(), hsum.data_ptr(), st), 'dist')
    torch.cuda.synchronize()
