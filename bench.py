#!/usr/bin/env python
"""Headline benchmark: VQ-VAE train step (BASELINE.json configs[1]: standard K=1024, 256x256, bs=32/GPU).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = preprocess -> encoder -> quantizer -> decoder -> MSE -> backward -> one flat-gradient all-reduce
-> AdamW, on a synthetic U(0,1) batch that is resident in HBM before the timed region.  Rank 0 prints ONE
JSON line.  `roofline` is for the dominant kernel (the implicit-GEMM conv): algorithmic FLOPs of its launches
(2*M*Cout*Cin*k*k each) over their HIP-event durations, recorded on the launch stream -- inside the timed region when the step
is issued eagerly (--no-graph), in eager steps of the same shapes right after it when the timed region replays hipGraphs (events
cannot be read back from inside a replayed graph; `roofline.event_pass` says which).
`cpu_baseline` times the CPU oracle (oracle/vqvae_oracle.py, a PyTorch-CPU restatement: kind "port") on a
bounded sample of the same workload, rank 0, N=1 only.
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIG_OF = {'standard': 'standard_vqvae.yaml', 'ema': 'ema_vqvae.yaml', 'entropy': 'entropy_vqvae.yaml',
             'gumbel': 'gumbel_vqgan.yaml'}
MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3, 'bf16x3': 2500.0}     # MI355X_MICROARCH.md, dense (bf16x3: fp32 storage, three bf16 products per multiply-add -- priced against the bf16 pipe it runs on)
HBM_PEAK_BPS = 8.0e12                                   # MI355X_MICROARCH.md: HBM3E peak (about 6.3e12 achievable)


def run_config(args, world: int):
    """(run dict, yaml path, overrides): the YAML through train.py's derivation rules (vqvae/train.py:55-103), with the
    per-GPU batch of the benchmark (cumulative_bs = batch * world -> lr = base_lr * sqrt(cumulative_bs / 256)) and
    BASELINE.json's overrides (codebook size; VQ-GAN: adversarial phase on from epoch 0)"""
    train_mod = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.train')
    path = args.config or os.path.join(ROOT, 'example_confs', CONFIG_OF[args.quantizer])
    conf = train_mod.get_model_conf(path)
    over = {'training.cumulative_bs': args.batch * world, 'image_size': args.image_size}
    if args.codebook is not None:
        over['quantizer.num_embeddings'] = args.codebook
    if 'loss' in conf and not args.gan:
        conf.pop('loss')                                         # plain VQ-VAE step on a VQ-GAN yaml (MSE criterion)
    if 'loss' in conf:
        over['loss.adversarial_params.start_epoch'] = 0          # discriminator + R1 exercised
    return train_mod.derive_run_config(conf, world, over), path, over


def _cpu_model() -> str:
    try:
        for line in open('/proc/cpuinfo'):
            if line.lower().startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def _mem_available_gb() -> float:
    try:
        for line in open('/proc/meminfo'):
            if line.startswith('MemAvailable'):
                return int(line.split()[1]) / 1048576.0
    except Exception:
        pass
    return 0.0


def cpu_baseline(run: dict, image_size: int, batch: int, steps: int):
    """CPU oracle train step (fwd + bwd + AdamW), fp32, all host cores.  Default: ONE timed step at the headline batch (32) after
    a warm-up step at batch 4 (the B = 32 workload itself, ~50 s); `--quick` / a box with too little free memory for
    the torch-CPU autograd graph of 32 images (measured 1.9 GB per image at 256x256; 2.2 budgeted) times batch 4 and says so."""
    from oracle import vqvae_oracle as O
    asked = batch
    need_gb = 2.2 * batch * (image_size / 256.0) ** 2 + 8.0
    while batch > 4 and _mem_available_gb() < need_gb:
        batch //= 2
        need_gb = 2.2 * batch * (image_size / 256.0) ** 2 + 8.0
    model_mod = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.model')
    torch.manual_seed(1234)
    m = model_mod.VQVAE(image_size, run['ae_conf'], run['q_conf'], None, run['t_conf'])   # CPU tensors: init only
    params = {k: v.detach().clone().contiguous() for k, v in m.state_dict().items()}
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:                                       # honour a cgroup CPU quota (containers often expose every host CPU)
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    cores = min(cores, 64)                     # torch-CPU conv stops scaling (and starts thrashing) well before that
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(1234)
    images = torch.rand(batch, 3, image_size, image_size, generator=g)
    decay, _ = O.decay_split(list(params))
    v = {k: torch.zeros_like(p) for k, p in params.items()}
    times = []
    first = None
    params0 = {k: v.clone() for k, v in params.items()}
    budget_t0 = time.perf_counter()
    for s in range(steps + 1):
        if s >= 2 and time.perf_counter() - budget_t0 > 90.0:      # keep the default run within minutes
            steps = s - 1
            break
        t0 = time.perf_counter()
        # the warm-up step (thread pool, allocator, oneDNN primitive cache) runs on 4 images when the timed batch is larger
        r = O.train_step_mse(images[:4] if (s == 0 and batch > 4) else images, params, run['ae_conf']['num_res_blocks'],
                             len(run['ae_conf']['channel_multipliers']), 'standard', dict(commitment_cost=0.25))
        if s == 0 and batch > 4:
            times.append(time.perf_counter() - t0)
            continue                                               # (no parameter update from the short warm-up: step 1 starts from params0)
        if first is None:
            first = r
        for k_, gr in r['grads'].items():
            params[k_], _, v[k_] = O.adamw_step(params[k_], gr, v[k_], s + 1, run['t_conf']['lr'], 0.0, 0.99, 1e-8,
                                                1e-4 if k_ in decay else 0.0)
        times.append(time.perf_counter() - t0)
    t = sum(times[1:]) / steps
    return dict(value=round(batch / t, 4), unit='images/sec', cores=cores, kind='port', cpu=_cpu_model(),
                # B = 32 (the default when the host has the memory), or a smaller batch with an explicit note --
                # images/s of a CPU conv step is flat in the batch size at these sizes
                batch=batch, extrapolated_from_batch=(None if batch == 32 else batch),
                note=(None if batch == asked else f'batch {asked} asked, {batch} timed: {_mem_available_gb():.0f} GB of host memory available'),
                sample=f'{steps} timed step(s) (+1 warm-up at batch {min(4, batch)}) of the same train step at batch {batch}, fp32, '
                       f'torch-CPU oracle on {cores} threads of {_cpu_model()}; {t:.2f} s/step'), (params0, images, first)


def parity_cost(run: dict, image_size: int, oracle_step, device) -> dict:
    """what the benchmarked precision costs in parity: the throughput (bf16) mode of THIS build on the inputs and weights of
    the CPU oracle's first step (fp32 restatement of the reference) -- index agreement, reconstruction error, loss, cosine
    of the whole-model gradient.  The fp32 parity mode holds indices bit-exact / recon 1e-5 on the same comparison
    (tests/test_gpu_fullsize.py); its speed is `other_configs['... fp32 parity mode']`."""
    model_mod = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.model')
    trainer_mod = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.trainer')
    params0, images, r = oracle_step
    out = {}
    for label, dt in (('bf16', torch.bfloat16), ('bf16x3', 'bf16x3'), ('f32', torch.float32)):
        m = model_mod.VQVAE(image_size, run['ae_conf'], run['q_conf'], None, run['t_conf'], compute_dtype=dt)
        m.load_state_dict(params0, strict=True)
        m = m.to(device).train()
        tr = trainer_mod.MiniTrainer(num_training_batches=1)
        opt = tr.attach(m)[0]
        opt.zero_grad()
        loss = m.training_step(images.to(device), 0)
        loss.backward()
        named = dict(m.named_parameters())
        num = da = db = 0.0
        for k, gr in r['grads'].items():
            a = named[k].grad.detach().double().cpu().reshape(-1)
            b = gr.double().reshape(-1)
            num += float((a * b).sum()); da += float((a * a).sum()); db += float((b * b).sum())
        with torch.no_grad():
            recon, _, idx = m(m.preprocess_batch(images.to(device)))
        out[label] = dict(indices_equal_pct=round(100.0 * float((idx.cpu().reshape(-1) == r['idx'].reshape(-1)).float().mean()), 2),
                          recon_rel_err=float(f"{float((recon.float().cpu() - r['recon']).norm() / r['recon'].norm()):.3e}"),
                          loss=round(float(loss.detach()), 6), oracle_loss=round(float(r['loss']), 6),
                          gradient_cosine=round(num / (da * db) ** 0.5, 6), gradient_norm_ratio=round((da / db) ** 0.5, 4))
        del m, tr, opt
    out['sample'] = f'batch {images.shape[0]} at {image_size}x{image_size}, random-init weights, the CPU oracle step of cpu_baseline'
    return out


OTHER_CONFIGS = {      # BASELINE.json configs[2..4], single-GPU part, at their per-GPU batch
    'ema_vqvae cb=1024 bs=32': ['--quantizer', 'ema', '--codebook', '1024', '--batch', '32'],
    'entropy_vqvae cb=8192 bs=64': ['--quantizer', 'entropy', '--codebook', '8192', '--batch', '64'],
    'gumbel_vqgan bs=16 (LPIPS + discriminator + R1)': ['--gan', '--batch', '16'],
    # the headline config again with ordered partial sums instead of atomics (vqvae/train.py:130: Trainer(deterministic=True))
    'standard_vqvae cb=1024 bs=32, deterministic mode': ['--deterministic'],
    # the mode in which north_star's parity statement holds bit for bit (fp32 storage, exact-fp32 MFMA): what parity costs
    'standard_vqvae cb=1024 bs=8, fp32 parity mode': ['--dtype', 'f32', '--batch', '8'],
    # the parity-GRADE mode on the bf16 matrix pipe (fp32 storage, every conv product as three bf16 products, fp32 accumulation:
    # indices / reconstructions / gradients within the fp32 tests' tolerances, tests/test_gpu_fullsize.py) at the headline batch
    'standard_vqvae cb=1024 bs=32, bf16x3 parity-grade mode': ['--dtype', 'bf16x3', '--batch', '32'],
}


def _child(argv, timeout=600, wrap=None):
    """this script again as a child process (its own HIP context), bench line parsed from stdout"""
    import subprocess
    env = dict(os.environ, VQK_BENCH_CHILD='1')
    cmd = (wrap or []) + [sys.executable, os.path.abspath(__file__)] + argv
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd='/tmp' if wrap else None)
    for line in reversed(r.stdout.splitlines()):
        if line.startswith('{'):
            return json.loads(line), r
    return None, r


def other_configs(steps: int = 20, warmup: int = 5) -> dict:
    """ms/step and images/s of the other BASELINE.json configs on this GPU: >= 20 timed steps each, same timing rules"""
    res = {}
    for label, argv in OTHER_CONFIGS.items():
        try:
            j, r = _child(argv + ['--steps', str(steps), '--warmup', str(warmup), '--no-cpu-baseline'])
            if j is None:
                res[label] = dict(error=(r.stderr or r.stdout)[-300:])
            else:
                res[label] = dict(ms_per_step=j['ms_per_step'], images_per_sec=j['value'], steps=j['steps'], dtype=j['dtype'],
                                  launch=j['config']['launch'], workload=j['config']['workload'])
                sr, rl = j.get('step_roofline'), j.get('roofline')
                if sr:                            # the whole step against the matrix-pipe peak of its dtype, and its dominant kernel
                    res[label].update(algorithmic_tflop=sr['algorithmic_tflop'], executed_tflop=sr['executed_tflop'],
                                      frac_of_peak=sr['frac_of_peak'], executed_frac_of_peak=sr['executed_frac_of_peak'],
                                      peak_tflops=sr['peak_tflops'], counted=sr['counted'])
                if rl:
                    res[label].update(dominant_kernel=rl['kernel'], dominant_kernel_avg_launch_us=rl['avg_kernel_launch_us'],
                                      dominant_kernel_frac=rl['frac'], dominant_kernel_executed_frac=rl['executed_frac'], dominant_kernel_ms_per_step=round(rl['kernel_time_frac_of_step'] * j['ms_per_step'], 3))
        except Exception as exc:
            res[label] = dict(error=f'{type(exc).__name__}: {exc}')
    return res


GN_FAMILIES = {      # bench event name -> kernel symbols behind it (csrc/norm.hip)
    'group_norm_fwd (HBM)': ('gn_stats_kernel', 'gn_apply_fin_kernel', 'gn_apply_kernel', 'gn_small_fwd_kernel'),
    'group_norm_bwd (HBM)': ('gn_bwd_reduce_kernel', 'gn_bwd_apply_kernel', 'gn_small_bwd_kernel', 'gn_cluster_bwd_kernel',
                             'gn_bwd_finish_kernel'),
}


def measure_traffic(kernel_sub: str, events_per_step: int, argv):
    """HBM bytes per bench event of the dominant kernel -- and per STEP of the GroupNorm kernel families -- measured now: two
    rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE -- separate passes, counters only, as MI355X_MICROARCH.md prescribes) over a
    2-step eager run of this same command; bytes = 2 * FETCH_SIZE + WRITE_SIZE (KiB units; gfx950 tallies wide coalesced reads
    at half).  (None, None, {}) when rocprofv3 is absent or a pass fails."""
    import shutil
    import sqlite3
    import tempfile
    prof = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if prof is None:
        return None, None, {}
    keep = []
    skip = {'--steps', '--warmup', '--sustain-s', '--traffic', '--cpu-batch', '--cpu-steps'}
    it = iter(argv)
    for a in it:
        if a in skip:
            next(it, None)
        elif a not in ('--no-graph', '--no-cpu-baseline', '--no-kernel-events', '--no-other-configs', '--quick'):
            keep.append(a)
    steps = 2
    total, fam = {}, {k: {} for k in GN_FAMILIES}
    try:
        for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = tempfile.mkdtemp(prefix=f'vqk_pmc_{ctr}_', dir='/tmp')
            env_tmp = os.environ.get('TMPDIR')
            os.environ['TMPDIR'] = '/tmp'
            try:
                _, r = _child(keep + ['--steps', str(steps), '--warmup', '1', '--no-graph', '--no-cpu-baseline', '--no-kernel-events'],
                              timeout=420, wrap=[prof, '--pmc', ctr, '-d', d, '-o', 'p', '--'])
            finally:
                if env_tmp is None:
                    os.environ.pop('TMPDIR', None)
                else:
                    os.environ['TMPDIR'] = env_tmp
            dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith('.db')]
            if not dbs:
                return None, None, {}
            db = sqlite3.connect(dbs[0])
            tables = [r_[0] for r_ in db.execute("select name from sqlite_master where type='table'")]
            t = lambda pfx: next(x for x in tables if x.startswith(pfx))
            kd, ks, pe, pi = t('rocpd_kernel_dispatch'), t('rocpd_info_kernel_symbol'), t('rocpd_pmc_event'), t('rocpd_info_pmc')
            cols = [r_[1] for r_ in db.execute(f'pragma table_info({ks})')]
            name_col = 'display_name' if 'display_name' in cols else 'kernel_name'
            base = (f'select sum(e.value), count(*) from {pe} e join {pi} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id '
                    f"join {ks} s on d.kernel_id = s.id where p.name = '{ctr}' and ")
            q = base + (f"s.{name_col} like '%{kernel_sub}%' "
                        f"and s.{name_col} not like '%, 1, 256>%'")          # (the NTAP = 1 instantiations are the 1x1 convs: their own line)
            tot, n = db.execute(q).fetchone()
            for fname, syms in GN_FAMILIES.items():
                cond = ' or '.join(f"s.{name_col} like '%{sym}%'" for sym in syms)
                ft, fn = db.execute(base + f'({cond})').fetchone()
                fam[fname][ctr] = (float(ft or 0.0), int(fn or 0))
            db.close()
            shutil.rmtree(d, ignore_errors=True)
            if not n:
                return None, None, {}
            total[ctr] = (float(tot), int(n))
        # (warmup 1 + 2 timed) eager steps were traced: every step launches the same kernels
        traced_steps = steps + 1
        kib = (2.0 * total['FETCH_SIZE'][0] + total['WRITE_SIZE'][0]) / traced_steps / events_per_step
        fam_bytes = {k: dict(hbm_bytes_per_step=int((2.0 * v['FETCH_SIZE'][0] + v['WRITE_SIZE'][0]) / traced_steps * 1024),
                             fetch_bytes_per_step=int(2.0 * v['FETCH_SIZE'][0] / traced_steps * 1024),
                             write_bytes_per_step=int(v['WRITE_SIZE'][0] / traced_steps * 1024),
                             kernel_launches_per_step=v['FETCH_SIZE'][1] // traced_steps)
                     for k, v in fam.items() if v.get('FETCH_SIZE', (0, 0))[1]}
        return int(kib * 1024), (f'measured in this run: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) over {traced_steps} '
                                 f'eager steps, {total["FETCH_SIZE"][1]} kernel launches; 2*FETCH_SIZE + WRITE_SIZE per bench event'), fam_bytes
    except Exception as exc:
        print(f'[bench] traffic pass failed ({type(exc).__name__}: {exc})', file=sys.stderr)
        return None, None, {}


# Reference box of the committed profile set (profiles/README.md lists every box of the round with these two figures): the
# normalised step time is what THIS run's step would take on that box if the MFMA-bound share of the step scaled with the
# conv-kernel calibration (share from profiles/round4_bench_kernel_stats.csv: the conv kernels are 0.70 of the step's critical
# path; GroupNorm / elementwise / AdamW 0.25 and the gaps 0.05 are taken as box-independent).
CALIB_REF = dict(conv_kernel_tflops=1253.0, mfma_loop_tflops=1145.9, hbm_copy_tbps=5.12)       # the box of profiles/round5_bench.json
CALIB_SHARES = dict(mfma=0.70, hbm=0.25)


def box_calibration(device, ms_per_step, seconds: float = 0.3) -> dict:
    """what THIS box delivers on two fixed kernels of the library (csrc/calib.hip), ~0.3 s each: the matrix-wave instruction
    mix of the role-split conv kernel on random bf16 operands, and a 1-GiB streaming copy.  Driver lines from different boxes
    become comparable: the same library ran 4-5 % apart across the pool in round 4."""
    native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')
    lib = native.lib()
    st = torch.cuda.current_stream().cuda_stream
    w = torch.empty(1 << 20, dtype=torch.uint8, device=device)
    sink = torch.zeros(4, dtype=torch.float32, device=device)
    native.check(lib.vqk_calib_fill(w.data_ptr(), w.numel(), st), 'calib_fill')
    iters, blocks = 256, 256
    flops = float(lib.vqk_calib_mfma_flops(iters, blocks))

    def timed(launch, per_batch):
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        total_ms, n = 0.0, 0
        t_host = time.perf_counter()
        while time.perf_counter() - t_host < seconds:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(per_batch):
                launch()
            e1.record()
            torch.cuda.synchronize()
            total_ms += e0.elapsed_time(e1)
            n += per_batch
        return total_ms * 1e-3 / n, n

    t_mfma, n_mfma = timed(lambda: native.check(lib.vqk_calib_mfma(w.data_ptr(), w.numel(), sink.data_ptr(), iters, blocks, st), 'calib_mfma'), 20)
    # ... and the dominant kernel ITSELF on its largest layer (128 -> 128 @256x256, 32 images: 618.5 GFLOP per launch): the
    # register-resident loop above turned out NOT to tell the boxes of this pool apart (1132-1150 TF on boxes whose steps differ
    # by 4 %) -- what differs is what a box sustains when the matrix pipe AND the memory system draw power
    ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
    gx = torch.Generator(device=device).manual_seed(7)
    cx = torch.randn(32, 128, 256, 256, device=device, generator=gx).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    cw = torch.randn(128, 3, 3, 128, device=device, generator=gx) * 0.03
    lay = ops.weight_layout(torch.bfloat16, 32, 256, 256, 128, 128, 3, False)
    cwq = ops.pack_weights(cw.reshape(-1), torch.bfloat16, 128, 128, 3, False, lay)
    t_conv, n_conv = timed(lambda: ops.raw_conv_fprop(cx, cwq, None, None, 3, False, 0, torch.bfloat16, 128, lay), 10)
    conv_flops = 2.0 * 32 * 256 * 256 * 128 * 128 * 9
    del cx
    nbytes = 1 << 30
    src = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dst = torch.empty(nbytes, dtype=torch.uint8, device=device)
    native.check(lib.vqk_calib_fill(src.data_ptr(), nbytes, st), 'calib_fill')
    t_copy, n_copy = timed(lambda: native.check(lib.vqk_calib_copy(src.data_ptr(), dst.data_ptr(), nbytes, st), 'calib_copy'), 5)
    del src, dst
    out = dict(conv_kernel_tflops=round(conv_flops / t_conv / 1e12, 1), conv_kernel_launches=n_conv,
               mfma_loop_tflops=round(flops / t_mfma / 1e12, 1), mfma_loop_launches=n_mfma,
               hbm_copy_tbps=round(2.0 * nbytes / t_copy / 1e12, 3), hbm_copy_launches=n_copy, seconds_each=seconds,
               kernels='conv3x3_mx_kernel on 128 -> 128 @256x256, 32 images, back to back; csrc/calib.hip: matrix-wave instruction mix of '
                       'that kernel on random bf16 operands, register / LDS / L2 resident (256 blocks x 4 waves); 1-GiB streaming copy '
                       '(read + write bytes)',
               reference_box=dict(CALIB_REF), shares=dict(CALIB_SHARES))
    if CALIB_REF['conv_kernel_tflops']:
        # only the conv-kernel figure enters: over the round's boxes it tracks the step (28.07 ms <-> 1252 TF, 28.39-28.49 <-> 1226-1228,
        # normalised 27.95-28.08), the copy figure does not (4.89-5.21 TB/s on boxes whose steps agree) and is reported only
        f = CALIB_SHARES['mfma'] * out['conv_kernel_tflops'] / CALIB_REF['conv_kernel_tflops'] + (1.0 - CALIB_SHARES['mfma'])
        out['ms_per_step_normalised'] = round(ms_per_step * f, 3)
        out['box_speed_vs_reference'] = round(f, 4)
    return out


def _flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def self_launch(n_gpus: int) -> int:
    """`python bench.py --gpus N` without a launcher: re-run this command line as N ranks (one per GPU of this node) under
    torch.distributed.run (vqvae/train.py:128-131 lets Lightning spawn the DDP ranks; here the bench does).  Returns the
    launcher's exit code; rank 0 of the child job prints the JSON line."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n_gpus:
        print(f'bench.py: --gpus {n_gpus} needs {n_gpus} visible GPUs, this box has {have}', file=sys.stderr)
        return 2
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
               OMP_NUM_THREADS=os.environ.get('OMP_NUM_THREADS', '4'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n_gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', type=str, default=None, help='example_confs-schema YAML (default: the one of --quantizer)')
    ap.add_argument('--batch', type=int, default=32, help='images per GPU')
    ap.add_argument('--image-size', type=int, default=256)
    ap.add_argument('--dtype', choices=['bf16', 'f32', 'bf16x3'], default='bf16')
    ap.add_argument('--quantizer', choices=['standard', 'ema', 'entropy', 'gumbel'], default='standard')
    ap.add_argument('--gan', action='store_true', help='VQ-GAN criterion of gumbel_vqgan.yaml (LPIPS + StyleGAN2 discriminator, non-saturating, R1 every 16 steps)')
    ap.add_argument('--codebook', type=int, default=None, help='override quantizer.num_embeddings of the YAML')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=32, help='batch of the timed CPU-oracle step (cpu_baseline); halved until it fits the host memory')
    ap.add_argument('--cpu-steps', type=int, default=1)
    ap.add_argument('--quick', action='store_true', help='cpu_baseline at batch 4 x 3 steps (extrapolated_from_batch = 4), as in rounds 1-5')
    ap.add_argument('--no-kernel-events', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='issue every kernel eagerly instead of replaying a hipGraph')
    ap.add_argument('--allow-eager', action='store_true',
                    help='multi-GPU runs: keep going (with "degraded" in the line) when the hipGraph capture fails; default: exit 3')
    ap.add_argument('--no-calibration', action='store_true', help='skip the 0.6 s box calibration (box_calibration in the line)')
    ap.add_argument('--deterministic', action='store_true', help='deterministic mode (vqvae/train.py:130): ordered partial sums instead of atomics; reports its cost')
    ap.add_argument('--sustain-s', type=float, default=10.0,
                    help='after the K timed steps keep stepping for this many seconds and report the sustained ms/step '
                         '(clock / power settle below the short run; 0 = off; N=1 only)')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip the short runs of the other BASELINE.json configs (ema, entropy K=8192 bs=64, gumbel VQ-GAN bs=16)')
    ap.add_argument('--traffic', choices=['auto', 'off'], default='auto',
                    help='auto: measure HBM bytes per launch of the dominant kernel with rocprofv3 --pmc passes of a 2-step '
                         'eager run (falls back to the committed profiles/ figure, labelled as such)')
    args = ap.parse_args()
    if os.environ.get('VQK_BENCH_CHILD') == '1':
        args.no_other_configs, args.traffic, args.sustain_s = True, 'off', 0.0
    if args.quick:
        args.cpu_batch, args.cpu_steps = 4, 3

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the train step is HIP kernels only (no CPU fallback)')
    if (args.gpus > 1 or os.environ.get('VQK_BENCH_SELF_LAUNCH') == '1') and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(args.gpus))          # plain `python bench.py --gpus N`: one rank per GPU, this node
    trainer_mod = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.trainer')
    model_mod = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.model')
    ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
    if hasattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch'):
        # the capture's settling steps run on the capture stream, the eager event pass on the default stream: intended
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
    # VQK_BENCH_ONE_GPU=1 (dry run of the multi-rank path on a 1-GPU box: `torchrun --nproc-per-node 2 bench.py --gpus 2` with every
    # rank on cuda:0 over gloo -- RCCL refuses two ranks per device): capture on several ranks at once, the form A/B, the barriers and
    # the MAX-over-ranks timing run as they will on a real node; the number it prints is NOT a scaling figure and says so
    one_gpu = os.environ.get('VQK_BENCH_ONE_GPU') == '1'
    if one_gpu:
        os.environ['LOCAL_RANK'] = '0'
    rank, local, world = trainer_mod.init_distributed('gloo' if one_gpu else 'nccl')
    if one_gpu:
        # ranks as PROCESSES on one GPU: no cluster form of the GroupNorm backward (its waiting blocks can be starved by the other
        # rank's launch: 602 / 868 timed-out blocks in a dry run, caught by check_kernel_health) -- ops.py: cluster_owner_ok
        native0 = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')
        native0.check(native0.lib().vqk_set_tuning(b'GN_CLUSTER_MAX_HW', 0), 'set_tuning')
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (the launcher started {world} ranks); '
                         f'pass --gpus {world} or launch --nproc-per-node {args.gpus}')
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    dtype = torch.bfloat16 if args.dtype == 'bf16' else 'bf16x3' if args.dtype == 'bf16x3' else torch.float32

    torch.manual_seed(1234)                                   # identical replicas on every rank
    if args.gan:
        args.quantizer = 'gumbel' if args.config is None else args.quantizer
    run, conf_path, conf_over = run_config(args, world)
    qtype, codebook = run['q_conf']['type'], run['q_conf']['num_embeddings']
    model = model_mod.VQVAE(run['image_size'], run['ae_conf'], run['q_conf'], run['l_conf'], run['t_conf'],
                            compute_dtype=dtype).to(device)
    args.gan = run['use_adversarial']
    if args.gan:
        model.criterion.discriminator.compute_dtype = model.compute_dtype
        model.criterion.perceptual_loss.net.compute_dtype = model.compute_dtype
    model.train()
    trainer = trainer_mod.MiniTrainer(num_training_batches=args.steps + args.warmup, deterministic=True if args.deterministic else None)
    trainer.attach(model)
    if os.environ.get('VQK_FORCE_DIST') == '1':
        for o in trainer.optimizers:
            o.force_collective = True
    model.on_train_start()
    g = torch.Generator().manual_seed(1234 + rank)
    images = torch.rand(args.batch, 3, args.image_size, args.image_size, generator=g).to(device)
    native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')

    def barrier():
        if world > 1 or dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = not args.no_graph      # (VQ-GAN: three graphs -- AE half, discriminator half, discriminator half with R1)
    step_fn = trainer.train_batch
    degraded = None
    comm_ab = None
    if use_graph:
        try:
            # Data parallel: which form of the gradient all-reduce is faster HERE is measured, not assumed -- the three ranges
            # reduced under the remaining backward (a collective's kernel holds CUs next to the persistent conv grids: those
            # draw their tiles from a queue then, TILE_QUEUE) or north_star's literal single flat all-reduce after the backward.
            # ~10 replayed steps each; the timed region runs in the faster form.  An explicit VQK_OVERLAP_ALLREDUCE /
            # VQK_TILE_QUEUE in the environment pins the form instead.
            pinned = 'VQK_OVERLAP_ALLREDUCE' in os.environ or 'VQK_TILE_QUEUE' in os.environ
            if dist.is_initialized() and not args.gan and not pinned:
                # (mode 1 = what init_distributed gives train.py users by default: first tile static, the rest drawn)
                forms = [('overlap_queue', True, 2), ('overlap_queue1', True, 1), ('overlap', True, 0), ('flat', False, 0)]
                comm_ab, trainers = {}, {}
                for name, overlap, tq in forms:
                    native.check(native.lib().vqk_set_tuning(b'TILE_QUEUE', tq), 'set_tuning')
                    t = trainer_mod.MiniTrainer(num_training_batches=args.steps + args.warmup, deterministic=True if args.deterministic else None)
                    t.optimizers, t.overlap_allreduce, model.trainer = trainer.optimizers, overlap, t
                    ok = 1
                    try:
                        t.capture(model, images, warmup=1)
                    except Exception as exc:                 # this FORM could not be captured here: the others still can be timed
                        ok = 0
                        print(f'[bench] rank {rank}: form {name} not captured ({type(exc).__name__}: {exc})', file=sys.stderr)
                        torch.cuda.synchronize()
                    agree = torch.tensor([ok], dtype=torch.int32, device=device)
                    dist.all_reduce(agree, op=dist.ReduceOp.MIN)          # every rank takes the same branch
                    if int(agree.item()) == 0:
                        comm_ab[name + '_ms'] = None
                        continue
                    for i in range(3):
                        t.train_batch_graphed(model, images, i)
                    barrier()
                    ta = time.perf_counter()
                    for i in range(10):
                        t.train_batch_graphed(model, images, i)
                    barrier()
                    dt_ab = torch.tensor([(time.perf_counter() - ta) / 10], dtype=torch.float64, device=device)
                    dist.all_reduce(dt_ab, op=dist.ReduceOp.MAX)
                    comm_ab[name + '_ms'] = round(float(dt_ab.item()) * 1e3, 3)
                    trainers[name] = (t, tq)
                timed = [f for f in forms if comm_ab[f[0] + '_ms'] is not None]
                if not timed:
                    raise RuntimeError('none of the three reduction forms could be captured')
                best = min(timed, key=lambda f: comm_ab[f[0] + '_ms'])[0]
                comm_ab['chosen'] = best
                comm_ab['tile_queue_mode'] = dict((f[0], f[2]) for f in forms)[best]
                trainer, tq = trainers[best]
                model.trainer = trainer
                native.check(native.lib().vqk_set_tuning(b'TILE_QUEUE', tq), 'set_tuning')     # (the eager event pass launches with it too)
                del trainers
            else:
                trainer.capture(model, images, warmup=max(1, min(3, args.warmup)))
            step_fn = trainer.train_batch_graphed
        except Exception as exc:
            # a capture problem must never produce a silent ~40 % slower "scaling curve": the line says so at top level, and a
            # multi-GPU run stops with exit code 3 unless --allow-eager
            degraded = f'eager launches (hipGraph capture failed: {type(exc).__name__}: {exc})'
            import traceback
            traceback.print_exc()
            print(f'[bench] {degraded}', file=sys.stderr)
            torch.cuda.synchronize()
            use_graph = False
            if world > 1 and not args.allow_eager:
                if dist.is_initialized():
                    dist.destroy_process_group()
                if rank == 0:
                    print(json.dumps(dict(metric='images/sec/node (256x256 bs=32/GPU) VQ-VAE train step', value=None, n_gpus=world,
                                          degraded=degraded, error='capture failed on a multi-GPU run (pass --allow-eager to time eager launches)')),
                          flush=True)
                raise SystemExit(3)
    for i in range(args.warmup):
        step_fn(model, images, i)
    barrier()
    if not use_graph and not args.no_kernel_events:
        ops.KERNEL_EVENTS = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step_fn(model, images, args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    ops.check_kernel_health()          # a timed-out GroupNorm cluster = wrong gradients: the line must not be printed as if nothing happened
    events, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    sustained = None
    if world == 1 and args.sustain_s > 0:
        # the K timed steps above are the metric; this is the same loop held for seconds, not fractions of one
        n_sus, t1 = 0, time.perf_counter()
        chunk = max(10, int(0.5 / max(elapsed / args.steps, 1e-4)))
        while time.perf_counter() - t1 < args.sustain_s:
            for i in range(chunk):
                step_fn(model, images, args.warmup + args.steps + n_sus + i)
            torch.cuda.synchronize()
            n_sus += chunk
        t_sus = time.perf_counter() - t1
        sustained = dict(seconds=round(t_sus, 2), steps=n_sus, ms_per_step=round(t_sus / n_sus * 1e3, 3),
                         images_per_sec=round(args.batch * n_sus / t_sus, 2))
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    event_steps = args.steps
    if use_graph and not args.no_kernel_events:
        # per-kernel HIP-event timing cannot be recorded inside a replayed graph: time the SAME kernels (same
        # shapes, same stream) in eager steps right after the timed region
        event_steps = 3
        ops.KERNEL_EVENTS = []
        overlap, ops.OVERLAP_WGRAD = ops.OVERLAP_WGRAD, False     # one kernel at a time: per-kernel durations, not overlap
        for i in range(event_steps):
            trainer.train_batch(model, images, args.warmup + args.steps + i)
        barrier()
        ops.OVERLAP_WGRAD = overlap
        events, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None

    calib = None
    if rank == 0 and not args.no_calibration and os.environ.get('VQK_BENCH_CHILD') != '1':
        try:
            calib = box_calibration(device, elapsed / args.steps * 1e3)
        except Exception as exc:
            calib = dict(error=f'{type(exc).__name__}: {exc}')
    if dist.is_initialized():
        dist.barrier()
    comm = None
    if dist.is_initialized():
        # proof of what RCCL saw, and what the collectives cost on the critical path (same loop, collectives muted)
        opt0 = trainer.optimizers[0]
        vqm = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.modules.vector_quantizers')
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)
        c0 = sum(o.collectives_issued for o in trainer.optimizers) + vqm.EMA_COLLECTIVES[0]
        b0 = sum(o.collective_bytes for o in trainer.optimizers) + vqm.EMA_COLLECTIVES[1]
        probe = max(5, min(args.steps, 20))
        barrier()
        ta = time.perf_counter()
        for i in range(probe):
            step_fn(model, images, args.warmup + args.steps + i)
        barrier()
        t_on = (time.perf_counter() - ta) / probe
        c1 = sum(o.collectives_issued for o in trainer.optimizers) + vqm.EMA_COLLECTIVES[0]
        b1 = sum(o.collective_bytes for o in trainer.optimizers) + vqm.EMA_COLLECTIVES[1]
        for o in trainer.optimizers:
            o.mute_collectives = True
        barrier()
        ta = time.perf_counter()
        for i in range(probe):
            step_fn(model, images, args.warmup + args.steps + probe + i)
        barrier()
        t_off = (time.perf_counter() - ta) / probe
        for o in trainer.optimizers:
            o.mute_collectives = False
        tt = torch.tensor([t_on, t_off], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        comm = dict(rccl_ranks=int(ones.item()), backend=dist.get_backend(), collectives_per_step=round((c1 - c0) / probe, 2),
                    collective_mb_per_step=round((b1 - b0) / probe / 1e6, 2),
                    ms_per_step_with_collectives=round(float(tt[0]) * 1e3, 3), ms_per_step_collectives_muted=round(float(tt[1]) * 1e3, 3),
                    exposed_comm_ms=round(float(tt[0] - tt[1]) * 1e3, 3), probe_steps=probe,
                    overlap='gradient ranges all-reduced under the remaining backward (VQK_OVERLAP_ALLREDUCE=0: one flat all-reduce)'
                            if (trainer.overlap_allreduce and getattr(trainer, '_graph2', None) is not None) else 'one flat all-reduce after the backward')
        if comm_ab is not None:
            comm.update(comm_ab)                # overlap_queue_ms / overlap_ms / flat_ms of the ~10-step A/B, and which one the timed region ran

    roofline = None
    if events:
        by_kernel = {}
        for name, flops, nbytes, e0, e1, nlaunch, xflops in events:
            rec = by_kernel.setdefault(name, [0, 0.0, 0.0, 0.0, 0, 0.0])
            rec[0] += 1
            rec[1] += flops
            rec[2] += e0.elapsed_time(e1) * 1e-3
            rec[3] += nbytes
            rec[4] += nlaunch
            rec[5] += xflops
        name, (count, flops, secs, nbytes, klaunches, xflops) = max(((k, v) for k, v in by_kernel.items() if v[1] > 0), key=lambda kv: kv[1][2])
        traffic, traffic_source, fam_traffic = None, None, {}
        if rank == 0 and world == 1 and args.traffic == 'auto':
            traffic, traffic_source, fam_traffic = measure_traffic(name.split('<')[0], count // event_steps, sys.argv[1:])
        if traffic is None:
            try:                               # HBM bytes per launch from the committed rocprofv3 --pmc passes
                for tag in ('round3', 'round2', 'round1'):
                    tp = os.path.join(ROOT, 'profiles', f'{tag}_traffic.json')
                    if os.path.exists(tp):
                        tj = json.load(open(tp))
                        if tj['kernel'].startswith(name.split('<')[0]):
                            traffic = tj['hbm_bytes_per_launch']
                            traffic_source = f'from profiles/{tag}_traffic.json (not measured in this run)'
                            break
            except Exception:
                pass
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        achieved = flops / secs / 1e12
        roofline = dict(bound='mfma', achieved=round(achieved, 2), peak=peak, unit='TFLOP/s',
                        frac=round(achieved / peak, 4), traffic=traffic, traffic_source=traffic_source, kernel=name,
                        # `achieved` counts ALGORITHMIC multiply-adds (an upsample conv in phase form = the 3x3 conv it replaces);
                        # the matrix pipe executes 4/9 of them for those events: hardware utilisation is the executed figure
                        executed_tflops=round(xflops / secs / 1e12, 2), executed_frac=round(xflops / secs / 1e12 / peak, 4),
                        algorithmic_bytes_per_launch=int(nbytes / count),
                        launches_per_step=count // event_steps, avg_launch_us=round(secs / count * 1e6, 2),
                        # an upsample conv in phase form is ONE event (one algorithmic 3x3 conv) but FOUR kernel launches:
                        # this is the figure that compares with rocprofv3's per-kernel average (profiles/*_kernel_stats.csv)
                        kernel_launches_per_step=klaunches // event_steps,
                        avg_kernel_launch_us=round(secs / klaunches * 1e6, 2),
                        avg_gflop_per_launch=round(flops / count / 1e9, 3),
                        kernel_time_frac_of_step=round((secs / event_steps) / (elapsed / args.steps), 3),
                        event_pass=('eager steps after the timed region, kernels serialised (no wgrad side stream)' if use_graph
                                    else 'timed region'),
                        all_kernels={k: (dict(launches=v[0] // event_steps, ms_per_step=round(v[2] / event_steps * 1e3, 3),
                                              tflops=round(v[1] / v[2] / 1e12, 1)) if v[1] > 0 else
                                         dict(launches=v[0] // event_steps, ms_per_step=round(v[2] / event_steps * 1e3, 3),
                                              algorithmic_tbps=round(v[3] / v[2] / 1e12, 2),
                                              hbm_frac=round(v[3] / v[2] / HBM_PEAK_BPS, 3)))
                                     for k, v in by_kernel.items()})
        # measured HBM bytes of the GroupNorm passes against their algorithmic bytes: > 1.1 would be wasted re-reads
        for k, fb in fam_traffic.items():
            if k in roofline['all_kernels'] and k in by_kernel and by_kernel[k][3] > 0:
                alg = by_kernel[k][3] / event_steps
                roofline['all_kernels'][k].update(measured_hbm_bytes_per_step=fb['hbm_bytes_per_step'],
                                                  algorithmic_bytes_per_step=int(alg),
                                                  traffic_over_algorithmic=round(fb['hbm_bytes_per_step'] / alg, 3),
                                                  measured_tbps=round(fb['hbm_bytes_per_step'] / (by_kernel[k][2] / event_steps) / 1e12, 2),
                                                  traffic_source='rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE of this run (2*FETCH + WRITE, eager steps)')
        if fam_traffic:
            roofline['gn_traffic'] = fam_traffic

    # the WHOLE step against the matrix-pipe peak: algorithmic FLOPs of every timed matrix-pipe event of one step (convs of the
    # autoencoder / LPIPS / discriminator, the quantizer's GEMM-shaped launches that go through the conv kernels) plus the
    # quantizer's distance evaluations (2 N K D each; standard / EMA / Gumbel lookup: one; entropy: forward + the backward's
    # recomputation, exact-fp32 MFMA, priced at the step's peak like everything else)
    step_roofline = None
    if events:
        nvec = args.batch * (args.image_size // 16) ** 2
        dist_evals = 2 if qtype == 'entropy' else 0 if qtype == 'gumbel' else 1      # (Gumbel: the code lookup is a timed GEMM event)
        dist_tflop = dist_evals * 2.0 * nvec * codebook * run['q_conf']['embedding_dim'] / 1e12
        alg = sum(e[1] for e in events) / event_steps / 1e12 + dist_tflop
        exe = sum(e[6] for e in events) / event_steps / 1e12 + dist_tflop
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        ms = elapsed / args.steps * 1e3
        step_roofline = dict(algorithmic_tflop=round(alg, 3), executed_tflop=round(exe, 3), peak_tflops=peak,
                             frac_of_peak=round(alg / (ms * 1e-3) / peak, 4), executed_frac_of_peak=round(exe / (ms * 1e-3) / peak, 4),
                             quantizer_distance_tflop=round(dist_tflop, 4),
                             counted='timed conv / GEMM events of one step + the quantizer distance evaluations; wall time of the replayed step')

    result_line = None
    vq_kernel = None
    if rank == 0 and not args.no_kernel_events:
        # the other kernel BASELINE.json's north_star names: the VQ distance / argmin kernel at this config's (N, K, D)
        try:
            sys.path.insert(0, os.path.join(ROOT, 'tools'))
            import vqbench
            vq_kernel = vqbench.bench(args.batch * (args.image_size // 16) ** 2, codebook, run['q_conf']['embedding_dim'], iters=200)
            vq_kernel['note'] = ('indices bit-exact vs the oracle; algorithmic bytes = z + codebook + q + idx (17.9 MB at '
                                 '(8192, 1024, 256): 2.2 us at the HBM peak, i.e. below one launch -- the kernel is bound by the '
                                 "matrix pipe and the L2 stream of the codebook, DESIGN.md 3)")
        except Exception as exc:
            vq_kernel = dict(error=f'{type(exc).__name__}: {exc}')
    if rank == 0:
        cpu = parity = None
        if world == 1 and not args.no_cpu_baseline:
            cpu, oracle_step = cpu_baseline(run, args.image_size, args.cpu_batch, args.cpu_steps)
            if qtype == 'standard' and not args.gan:
                try:
                    parity = parity_cost(run, args.image_size, oracle_step, device)
                except Exception as exc:
                    parity = dict(error=f'{type(exc).__name__}: {exc}')
        value = world * args.batch * args.steps / elapsed
        out = dict(metric='images/sec/node (256x256 bs=32/GPU) VQ-VAE train step', value=round(value, 2),
                   unit='images/sec', n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(elapsed / args.steps * 1e3, 3), higher_is_better=True, scaling='weak',
                   vs_baseline=None, dtype=args.dtype, data='synthetic',
                   config=dict(workload=f'{qtype}_{"vqgan" if args.gan else "vqvae"} cb={codebook}, '
                                        f'{args.image_size}x{args.image_size} bs={args.batch}/GPU (encoder+VQ+decoder '
                                        f'{"+LPIPS+discriminator+R1 " if args.gan else ""}fwd/bwd + AdamW)',
                               yaml=os.path.relpath(conf_path, ROOT), yaml_overrides=conf_over,
                               learning_rate=run['learning_rate'],
                               global_batch=world * args.batch, parallelism=f'dp{world}',
                               launch=(('three hipGraphs (fwd + decoder bwd | quantizer + deep encoder bwd | encoder head bwd), decoder-range '
                                        'all-reduce under the second, deep-encoder range under the third, head all-reduce, AdamW')
                                       if (use_graph and getattr(trainer, '_graph3', None) is not None) else
                                       ('two hipGraphs (fwd + decoder bwd | quantizer + encoder bwd), decoder-range all-reduce under the second, '
                                        'tail all-reduce, AdamW') if (use_graph and getattr(trainer, '_graph2', None) is not None) else
                                       'three hipGraphs (AE half | discriminator half | discriminator half + R1), optimizer steps between' if (use_graph and args.gan) else
                                       'hipGraph replay (fwd+bwd) + eager all-reduce + AdamW' if use_graph else 'eager'),
                               deterministic=bool(ops.DETERMINISTIC), final_loss=round(float(loss.item()), 6),
                               rccl_ranks=None if comm is None else comm['rccl_ranks'],
                               collectives_per_step=0 if comm is None else comm['collectives_per_step']),
                   roofline=roofline, step_roofline=step_roofline, cpu_baseline=cpu, vq_kernel=vq_kernel, sustained=sustained, comm=comm,
                   bf16_vs_fp32_oracle=parity, box_calibration=calib)
        if degraded is not None:
            out['degraded'] = degraded
        if one_gpu and world > 1:
            out['degraded'] = f'dry run: {world} ranks share ONE GPU over gloo (VQK_BENCH_ONE_GPU=1) -- control flow only, not a scaling figure'
        if world == 1 and not args.no_other_configs:
            out['other_configs'] = other_configs()
        result_line = json.dumps(out)
    if dist.is_initialized():
        dist.destroy_process_group()
    _flush_c_stdio()                   # RCCL's version banner sits in the C stdio buffer: push it out BEFORE the JSON line
    if result_line is not None:
        print(result_line, flush=True)


if __name__ == '__main__':
    main()
