import importlib, os, sys, torch, collections, traceback
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, R)
import bench, argparse
trainer_mod = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.trainer')
model_mod = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.model')
ns = argparse.Namespace(config=None, quantizer='standard', batch=8, image_size=256, codebook=None, gan=False)
run, _, _ = bench.run_config(ns, 1)
torch.manual_seed(0)
m = model_mod.VQVAE(256, run['ae_conf'], run['q_conf'], None, run['t_conf'], compute_dtype=torch.bfloat16).cuda().train()
tr = trainer_mod.MiniTrainer(num_training_batches=10); tr.attach(m); m.on_train_start()
x = torch.rand(8, 3, 256, 256).cuda()
for i in range(2): tr.train_batch(m, x, i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_batch(m, x, 2)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    n = ev.name
    if 'copy' in n.lower() or 'fill' in n.lower() or 'zero' in n.lower() or n in ('aten::add', 'aten::add_', 'aten::clone', 'aten::contiguous', 'aten::to', 'aten::_to_copy'):
        st = [f for f in (ev.stack or []) if 'vqvae-vqgan' in f or 'bench' in f]
        cnt[(n, st[0] if st else '?')] += 1
for (n, st), c in cnt.most_common(45):
    print(c, n, st[-110:])
