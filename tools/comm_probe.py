#!/usr/bin/env python
"""What does a collective's kernel running BESIDE the backward cost on one GPU?  (the one-GPU world-1
all-reduce launches no ring kernel and moves no peer traffic, so it says nothing about N > 1.)

The headline step is captured as usual (three hipGraphs with the gradient all-reduces issued between them,
trainer.MiniTrainer.train_batch_graphed).  Here every `all_reduce_range` is replaced by a STAND-IN for RCCL's kernel: `blocks`
persistent 256-thread blocks on a side stream that stream-add the range (bytes x 2 (N-1)/N of a ring all-reduce) at a throttled
rate, so that the chosen number of CUs stays occupied for the time the collective would take at a given bus bandwidth.  Printed,
per tile-assignment mode of the persistent conv kernels (tuning slot TILE_QUEUE: 0 static share, 1 / 2 dynamic queue): ms per
step without collectives and with the stand-in.  (Round 4 also measured smaller persistent grids -- tuning slot COMM_CUS --
and found them harmful: profiles/round4_comm_probe.txt.)  Usage: python tools/comm_probe.py [--steps 30] [--modes 0,1,2]"""
import argparse, importlib, os, sys, time
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ.setdefault('VQK_FORCE_DIST', '1')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29531')
import bench
trainer_mod = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.trainer')
model_mod = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.model')
native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')
lib = native.lib()


class Work:
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--modes', type=str, default='0,1,2', help='TILE_QUEUE modes to compare')
    ap.add_argument('--blocks', type=str, default='16,32,64', help='CUs held by the stand-in')
    ap.add_argument('--streams', type=int, default=8,
                    help='candidate streams for the stand-in: HIP streams that share a HARDWARE queue run one after the other (AQL '
                         'barrier bits), and a replayed hipGraph spreads over every hardware queue of the process (4 by default, '
                         'GPU_MAX_HW_QUEUES) -- the stand-in is measured on each candidate and kept on the one that overlaps best')
    ap.add_argument('--ms', type=str, default='0.5,1.5', help='minimum duration of the largest range (ms)')
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    trainer_mod.init_distributed('nccl')
    ns = argparse.Namespace(config=None, quantizer='standard', batch=32, image_size=256, codebook=None, gan=False)
    run, _, _ = bench.run_config(ns, 1)
    torch.manual_seed(1234)
    model = model_mod.VQVAE(256, run['ae_conf'], run['q_conf'], None, run['t_conf'], compute_dtype=torch.bfloat16).to(dev).train()
    tr = trainer_mod.MiniTrainer(num_training_batches=1000)
    opt = tr.attach(model)[0]
    opt.force_collective = True
    model.on_train_start()
    images = torch.rand(32, 3, 256, 256, generator=torch.Generator().manual_seed(1)).to(dev)
    tr.capture(model, images, warmup=2)
    sides = [torch.cuda.Stream() for _ in range(max(1, a.streams))]
    side_of = dict(s=sides[0])
    src = torch.zeros(opt.flat_g.numel(), device=dev)
    dst = torch.zeros(opt.flat_g.numel(), device=dev)
    state = dict(mode='off')

    def fake_range(lo, hi, async_op=True):
        opt.grad_scale = 1.0
        if state['mode'] == 'off' or hi <= lo:
            return None
        nbytes = (hi - lo) * 4
        nbytes -= nbytes % 16
        cur = torch.cuda.current_stream()
        side = side_of['s']
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            native.check(lib.vqk_probe_stream_add(src.data_ptr(), dst.data_ptr(), nbytes, state['blocks'], state['passes'], state['sleep'],
                                                  side.cuda_stream), 'probe')
            ev = torch.cuda.Event(); ev.record(side)
        return Work(ev)
    opt.all_reduce_range = fake_range

    cur = dict(tr=tr)

    def steps(n):
        t_ = cur['tr']
        for i in range(5):
            t_.train_batch_graphed(model, images, i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            t_.train_batch_graphed(model, images, i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def kernel_ms(nbytes, blocks, passes, sleep):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        lib.vqk_probe_stream_add(src.data_ptr(), dst.data_ptr(), nbytes, blocks, passes, sleep, torch.cuda.current_stream().cuda_stream)
        e0.record()
        lib.vqk_probe_stream_add(src.data_ptr(), dst.data_ptr(), nbytes, blocks, passes, sleep, torch.cuda.current_stream().cuda_stream)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    ranges = tr._ranges(opt, 3 if getattr(tr, '_graph3', None) is not None else 2)
    mb = [(hi - lo) * 4 / 1e6 for lo, hi in ranges]
    print(f'arena ranges (MB): {[round(m, 1) for m in mb]}  -- all-reduced after graph 1 / 2 / 3', flush=True)
    big = int(max(hi - lo for lo, hi in ranges)) * 4
    big -= big % 16
    # target: the LARGEST range takes `ms` (bus bandwidth = 1.75 x bytes / ms for an 8-GPU ring); the throttle is found by search
    cases = []
    for blocks in [int(b) for b in a.blocks.split(',')]:
        for ms in [float(m) for m in a.ms.split(',')]:
            sleep, passes = 0, 1
            t = kernel_ms(big, blocks, passes, sleep)
            while t < ms and sleep < 64:
                sleep = max(1, sleep * 2)
                t = kernel_ms(big, blocks, passes, sleep)
            cases.append((blocks, passes, sleep, t))
    # the tile assignment of the persistent conv kernels is part of the captured graphs: one capture per TILE_QUEUE mode.
    # Which candidate stream overlaps with the replayed graphs is found per capture (a capture's internal branches get their
    # hardware queues anew): the same case on each stream; a stream that shares the hardware queue of the graphs' main branch adds
    # the stand-in's whole duration to the step, one that shares a side branch's queue stalls that branch.
    names = {0: 'static share', 1: 'queue, first tile static', 2: 'queue, every tile'}
    for mode in [int(m) for m in a.modes.split(',')]:
        lib.vqk_set_tuning(b'TILE_QUEUE', mode)
        tr2 = trainer_mod.MiniTrainer(num_training_batches=1000)
        tr2.optimizers = tr.optimizers
        model.trainer = tr2
        tr2.capture(model, images, warmup=1)
        cur['tr'] = tr2
        state['mode'] = 'off'
        base = steps(a.steps)
        print(f'TILE_QUEUE={mode} ({names[mode]}): no collectives {base:.3f} ms/step', flush=True)
        blocks, passes, sleep, t = cases[0]
        per_stream = []
        for sd in sides:
            side_of['s'] = sd
            state.update(mode='on', blocks=blocks, passes=passes, sleep=sleep)
            per_stream.append(steps(10) - base)
        best = min(range(len(sides)), key=lambda i: per_stream[i])
        side_of['s'] = sides[best]
        print(f'  stand-in ({blocks} CUs, {t:.2f} ms) on {len(sides)} candidate streams: step +' + ' / +'.join(f'{d:.2f}' for d in per_stream) +
              f' ms -> stream #{best} (GPU_MAX_HW_QUEUES={os.environ.get("GPU_MAX_HW_QUEUES", "default 4")})', flush=True)
        for blocks, passes, sleep, t in cases:
            state.update(mode='on', blocks=blocks, passes=passes, sleep=sleep)
            on = steps(a.steps)
            print(f'  stand-in on {blocks:3d} CUs, largest range {t:.2f} ms (throttle {sleep}): {on:.3f} ms/step (+{on - base:.3f})', flush=True)
    lib.vqk_reset_tuning()


if __name__ == '__main__':
    main()
