"""forward-kernel timing only (ablation builds: VQK_LIB=scratch/libvqk_<tag>.so): fused forward at (8192, 1024, 256) and (4096, 1024, 256)"""
import importlib, os, sys, torch
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
import vqbench
native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')
lib = native.lib()
for n in (4096, 8192):
    k, d = 1024, 256
    g = torch.Generator().manual_seed(1)
    z = (torch.randn(n, d, generator=g) * 0.36).cuda()
    e = (z[torch.randperm(n, generator=g)[:k].cuda()] + 0.01 * torch.randn(k, d).cuda()).contiguous()
    ws = torch.empty(lib.vqk_vq_filter_ws_bytes(k, d), dtype=torch.uint8, device='cuda')
    s = torch.cuda.current_stream().cuda_stream
    lib.vqk_vq_prepare_f32(e.data_ptr(), k, d, ws.data_ptr(), ws.numel(), s)
    idx = torch.empty(n, dtype=torch.int64, device='cuda'); qlo = torch.empty(n, d, dtype=torch.bfloat16, device='cuda')
    sse = torch.zeros((), device='cuda'); hist = torch.zeros(k, dtype=torch.int32, device='cuda')
    fwd = lambda: lib.vqk_vq_forward_f32(z.data_ptr(), e.data_ptr(), ws.data_ptr(), ws.numel(), n, k, d, 0, idx.data_ptr(), 0, qlo.data_ptr(), sse.data_ptr(), hist.data_ptr(), s)
    asg = lambda: lib.vqk_vq_forward_f32(z.data_ptr(), e.data_ptr(), ws.data_ptr(), ws.numel(), n, k, d, 0, idx.data_ptr(), 0, 0, 0, 0, s)
    print(n, 'forward', round(vqbench._time(fwd, 300) * 1e6, 2), 'assign only', round(vqbench._time(asg, 300) * 1e6, 2))
