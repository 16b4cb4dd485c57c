#!/usr/bin/env python
"""Feasibility probe: the 32-image step as NSPLIT independent chains (forward + backward each), against the ordinary one-chain graph.
Timing only (the probe does not check the numerics).  Default: the chains one after the other on ONE stream, eager and captured
(round 4: 2 chains 32.6 ms against 28.2 ms for one -- half the tiles per launch, twice the launches -- so interleaving them on two
streams would have to hide more than 4.4 ms).  TWO_STREAMS=1: one stream per chain; the eager form runs, capturing it died inside
hipStreamEndCapture on ROCm 7.2 (segmentation fault), which is why this is opt-in."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], '--steps', '4', '--warmup', '2', '--no-cpu-baseline', '--no-kernel-events', '--no-other-configs', '--traffic', 'off',
            '--sustain-s', '0']
import bench


def main():
    hook = {}
    trainer_mod = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.trainer')
    ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
    orig = trainer_mod.MiniTrainer.train_batch_graphed

    def spy(self, model, batch, batch_index):
        hook['t'], hook['m'], hook['b'] = self, model, batch
        return orig(self, model, batch, batch_index)
    trainer_mod.MiniTrainer.train_batch_graphed = spy
    bench.main()
    tr, m, x = hook['t'], hook['m'], hook['b']
    opt = tr.optimizers[0]

    def t(fn, reps=20):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    print('one chain  (graph replay only) %.3f ms' % t(tr._graph.replay), flush=True)
    nsplit = int(os.environ.get('NSPLIT', '2'))
    xs = [c.clone() for c in x.chunk(nsplit, 0)]
    cap = torch.cuda.Stream()
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    if not os.environ.get('TWO_STREAMS'):
        streams = [cap] * nsplit
    m.defer_usage_accumulation = True

    def step():
        opt.zero_grad()
        cur = torch.cuda.current_stream()
        for xi, st in zip(xs, streams):
            if st is not cur: st.wait_stream(cur)
            with torch.cuda.stream(st):
                ops._stream()
                loss = m.training_step(xi, 0)
                (loss / nsplit).backward()
        for st in streams:
            if st is not cur: cur.wait_stream(st)
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(cap)
    torch.cuda.synchronize()
    print('%d chains, eager                %.3f ms' % (nsplit, t(step, 5)), flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap, capture_error_mode='thread_local'):
        step()
    torch.cuda.synchronize()
    print('%d chains  (graph replay only) %.3f ms' % (nsplit, t(g.replay)))


if __name__ == '__main__':
    main()
