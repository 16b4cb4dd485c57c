#!/usr/bin/env python
"""VQ assignment micro-benchmark: (N,K) in {(8192,1024),(16384,8192),(4096,1024)}, D=256, >=200 launches;
algorithmic bytes N*D*4 + K*D*4 + N*D*4 + N*8 (z read, codebook read, q write, idx write) over the assign kernel time."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')


def bench(n, k, d=256, iters=200, scale=0.36):
    g = torch.Generator().manual_seed(1234)
    z = (torch.randn(n, d, generator=g) * scale).cuda()
    e = ((torch.rand(k, d, generator=g) * 2 - 1) / k).cuda()
    lib = native.lib()
    z2 = torch.empty(n, device='cuda'); e2 = torch.empty(k, device='cuda')
    idx = torch.empty(n, dtype=torch.int64, device='cuda')
    s = torch.cuda.current_stream().cuda_stream
    lib.vqk_row_sqnorm_f32(z.data_ptr(), n, d, z2.data_ptr(), s)
    lib.vqk_row_sqnorm_f32(e.data_ptr(), k, d, e2.data_ptr(), s)
    fn = lambda: lib.vqk_vq_assign_f32(z.data_ptr(), e.data_ptr(), z2.data_ptr(), e2.data_ptr(), n, k, d, 0, idx.data_ptr(), s)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    nbytes = n * d * 4 + k * d * 4 + n * d * 4 + n * 8
    flops = 2.0 * n * k * d
    return dict(n=n, k=k, d=d, us=round(t * 1e6, 2), gbps=round(nbytes / t / 1e9, 1), hbm_frac=round(nbytes / t / 8e12, 4),
                tflops_fp32=round(flops / t / 1e12, 1), fp32_mfma_frac=round(flops / t / 157.3e12, 3))


if __name__ == '__main__':
    for n, k in [(8192, 1024), (16384, 8192), (4096, 1024)]:
        print(bench(n, k))
