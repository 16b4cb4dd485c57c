#!/usr/bin/env python
"""Per-layer-shape table of one train step: every timed kernel event of bench.py's event pass, keyed by (kernel, layer shape)
(VQK_EVENT_SHAPES=1), sorted by time.  usage: tools/per_shape.py [bench.py args...] > profiles/roundN_per_shape.txt"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, VQK_EVENT_SHAPES='1', VQK_BENCH_CHILD='1')
cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline', '--steps', '20', '--warmup', '5'] + sys.argv[1:]
r = subprocess.run(cmd, env=env, capture_output=True, text=True)
line = next((l for l in reversed(r.stdout.splitlines()) if l.startswith('{')), None)
if line is None:
    sys.exit(r.stderr[-2000:])
j = json.loads(line)
ak = j['roofline']['all_kernels']
print(f"# {j['config']['workload']}: {j['ms_per_step']} ms/step ({j['config']['launch']}); event pass: {j['roofline']['event_pass']}")
print(f"# {'kernel / layer shape':84s} {'n/step':>6s} {'us/evt':>9s} {'ms/step':>8s} {'TFLOP/s':>8s} {'of peak':>7s} {'TB/s':>6s}")
tot = 0.0
fam = {}
for name, v in sorted(ak.items(), key=lambda kv: -kv[1]['ms_per_step']):
    tf = v.get('tflops')
    tb = v.get('algorithmic_tbps')
    n = max(v['launches'], 1)
    print(f"{name:86s} {v['launches']:6d} {v['ms_per_step'] / n * 1e3:9.1f} {v['ms_per_step']:8.3f} "
          f"{(f'{tf:8.1f}' if tf else '        ')} {(f'{tf / 2500:7.3f}' if tf else '       ')} {(f'{tb:6.2f}' if tb else '')}")
    tot += v['ms_per_step']
    f = fam.setdefault(name.split(' ')[0], [0.0, 0.0])
    f[0] += v['ms_per_step']
    f[1] += (tf or 0.0) * v['ms_per_step']
print(f"# sum of timed events: {tot:.3f} ms/step (serialised)")
for k, (ms, tfms) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print(f"#   {k:50s} {ms:8.3f} ms/step" + (f"  {tfms / ms:8.1f} TFLOP/s = {tfms / ms / 2500:.3f} of peak" if tfms else ''))
