#!/usr/bin/env python
"""VQ assignment micro-benchmark: (N,K) in {(8192,1024),(16384,8192),(4096,1024)}, D=256, >=200 launches;
algorithmic bytes N*D*4 + K*D*4 + N*D*4 + N*8 (z read, codebook read, q write, idx write) over the assign kernel time."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')


def _time(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def bench(n, k, d=256, iters=200, scale=0.36, trained=True):
    """the assignment kernel the product runs (bf16 filter + exact re-rank, incl. its codebook-prep launch) and the
    exact-fp32 MFMA kernel it replaced, same inputs: a trained-like codebook (codes = perturbed latents) by default"""
    g = torch.Generator().manual_seed(1234)
    z = (torch.randn(n, d, generator=g) * scale)
    if trained and n >= k:
        e = z[torch.randperm(n, generator=g)[:k]] + 0.01 * torch.randn(k, d, generator=g)
    else:
        e = ((torch.rand(k, d, generator=g) * 2 - 1) / k)
    z, e = z.cuda(), e.cuda()
    lib = native.lib()
    z2 = torch.empty(n, device='cuda'); e2 = torch.empty(k, device='cuda')
    idx = torch.empty(n, dtype=torch.int64, device='cuda')
    idx2 = torch.empty(n, dtype=torch.int64, device='cuda')
    s = torch.cuda.current_stream().cuda_stream
    lib.vqk_row_sqnorm_f32(z.data_ptr(), n, d, z2.data_ptr(), s)
    lib.vqk_row_sqnorm_f32(e.data_ptr(), k, d, e2.data_ptr(), s)
    exact = lambda: lib.vqk_vq_assign_f32(z.data_ptr(), e.data_ptr(), z2.data_ptr(), e2.data_ptr(), n, k, d, 0, idx.data_ptr(), s)
    t_exact = _time(exact, iters)
    nbytes = n * d * 4 + k * d * 4 + n * d * 4 + n * 8
    flops = 2.0 * n * k * d
    out = dict(n=n, k=k, d=d, codebook='trained-like' if (trained and n >= k) else 'uniform')
    if d == 256 and k % 32 == 0 and ops.VQ_FILTER:
        ws = torch.empty(lib.vqk_vq_filter_ws_bytes(k, d), dtype=torch.uint8, device='cuda')
        filt = lambda: lib.vqk_vq_assign_filtered_f32(z.data_ptr(), e.data_ptr(), z2.data_ptr(), e2.data_ptr(), n, k, d, 0,
                                                      idx2.data_ptr(), ws.data_ptr(), ws.numel(), s)
        t = _time(filt, iters)
        out.update(kernel='vq_assign_filter_kernel (bf16 MFMA filter + exact fp32 re-rank; + vq_filter_prep_kernel)',
                   us=round(t * 1e6, 2), gbps=round(nbytes / t / 1e9, 1), hbm_frac=round(nbytes / t / 8e12, 4),
                   tflops_bf16=round(flops / t / 1e12, 1), indices_equal_exact_kernel=bool(torch.equal(idx, idx2)),
                   exact_fp32_kernel_us=round(t_exact * 1e6, 2))
        # round 4: the whole quantizer forward as ONE kernel (codebook derivatives prepared when the codebook changes) and the
        # whole backward as one -- against the launch sequence they replace
        lib.vqk_vq_prepare_f32(e.data_ptr(), k, d, ws.data_ptr(), ws.numel(), s)
        q32 = torch.empty(n, d, device='cuda'); qlo = torch.empty(n, d, dtype=torch.bfloat16, device='cuda')
        sse = torch.zeros((), device='cuda'); hist = torch.zeros(k, dtype=torch.int32, device='cuda')
        idx3 = torch.empty(n, dtype=torch.int64, device='cuda')
        fwd = lambda: lib.vqk_vq_forward_f32(z.data_ptr(), e.data_ptr(), ws.data_ptr(), ws.numel(), n, k, d, 0, idx3.data_ptr(), 0,
                                             qlo.data_ptr(), sse.data_ptr(), hist.data_ptr(), s)
        t_fwd = _time(fwd, iters)

        def old_fwd():
            lib.vqk_row_sqnorm_f32(z.data_ptr(), n, d, z2.data_ptr(), s)
            lib.vqk_row_sqnorm_f32(e.data_ptr(), k, d, e2.data_ptr(), s)
            filt()
            lib.vqk_vq_gather_f32(z.data_ptr(), e.data_ptr(), idx2.data_ptr(), n, k, d, q32.data_ptr(), qlo.data_ptr(), sse.data_ptr(),
                                  hist.data_ptr(), s)
        t_old_fwd = _time(old_fwd, iters)
        dq = torch.randn(n, d, device='cuda').to(torch.bfloat16)
        dz = torch.empty(n, d, device='cuda'); de = torch.zeros(k, d, device='cuda'); gs = torch.ones((), device='cuda')
        bw = lambda f: (lambda: f(z.data_ptr(), e.data_ptr(), idx2.data_ptr(), dq.data_ptr(), 1, n, k, d, 1e-7, 4e-7, gs.data_ptr(),
                                  dz.data_ptr(), de.data_ptr(), s))
        t_bwd, t_old_bwd = _time(bw(lib.vqk_vq_backward_fused_f32), iters), _time(bw(lib.vqk_vq_backward_f32), iters)
        fwd_bytes = n * d * 4 + k * d * 2 + n * d * 2 + n * 8          # z read, bf16 codebook read, bf16 q written, idx written
        out.update(forward_kernel='vq_assign_filter_kernel, fused form (|z|^2 + filter + re-rank + gather + loss sum + histogram)',
                   forward_us=round(t_fwd * 1e6, 2), forward_equal=bool(torch.equal(idx3, idx)),
                   forward_gbps=round(fwd_bytes / t_fwd / 1e9, 1), forward_hbm_frac=round(fwd_bytes / t_fwd / 8e12, 4),
                   forward_round3_sequence_us=round(t_old_fwd * 1e6, 2),
                   backward_us=round(t_bwd * 1e6, 2), backward_round3_sequence_us=round(t_old_bwd * 1e6, 2))
    else:
        out.update(kernel='vq_assign_reg_kernel (exact fp32 MFMA)', us=round(t_exact * 1e6, 2), gbps=round(nbytes / t_exact / 1e9, 1),
                   hbm_frac=round(nbytes / t_exact / 8e12, 4), tflops_fp32=round(flops / t_exact / 1e12, 1),
                   fp32_mfma_frac=round(flops / t_exact / 157.3e12, 3))
    return out


if __name__ == '__main__':
    for n, k in [(8192, 1024), (16384, 8192), (4096, 1024)]:
        print(bench(n, k))
    print(bench(8192, 1024, trained=False))
