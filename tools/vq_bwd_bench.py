import importlib, os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tools'))
import vqbench
native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')
lib = native.lib()
for n in (1024, 4096, 8192, 16384):
    k, d = 1024, 256
    g = torch.Generator().manual_seed(1)
    z = torch.randn(n, d, generator=g).cuda() * 0.36
    e = z[torch.randperm(n, generator=g)[:k].cuda()].contiguous() if n >= k else torch.randn(k, d).cuda()
    idx = torch.randint(0, k, (n,), generator=g).cuda()
    idx_c = torch.zeros(n, dtype=torch.int64).cuda()
    dq = torch.randn(n, d).cuda().to(torch.bfloat16)
    dz = torch.empty(n, d).cuda(); de = torch.zeros(k, d).cuda(); gs = torch.ones(()).cuda()
    s = torch.cuda.current_stream().cuda_stream
    f = lambda fn, ix, dep: (lambda: fn(z.data_ptr(), e.data_ptr(), ix.data_ptr(), dq.data_ptr(), 1, n, k, d, 1e-7, 4e-7, gs.data_ptr(), dz.data_ptr(), dep, s))
    print(n, 'fused', round(vqbench._time(f(lib.vqk_vq_backward_fused_f32, idx, de.data_ptr()), 200) * 1e6, 2),
          'fused dz only', round(vqbench._time(f(lib.vqk_vq_backward_fused_f32, idx, 0), 200) * 1e6, 2),
          'fused collapsed', round(vqbench._time(f(lib.vqk_vq_backward_fused_f32, idx_c, de.data_ptr()), 200) * 1e6, 2),
          'old', round(vqbench._time(f(lib.vqk_vq_backward_f32, idx, de.data_ptr()), 200) * 1e6, 2),
          'old dz only', round(vqbench._time(f(lib.vqk_vq_backward_f32, idx, 0), 200) * 1e6, 2),
          'old collapsed', round(vqbench._time(f(lib.vqk_vq_backward_f32, idx_c, de.data_ptr()), 200) * 1e6, 2))
