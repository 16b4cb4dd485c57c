#!/bin/bash
# Round-6 profile set, ONE gpurun call (one box).  -> gpurun_out/round6_*  (copied to profiles/ after a look)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/round6_smoke.txt 2>&1
tools/collect_profiles.sh round6 conv3x3_mx
python - <<'PY'
import json
j = json.loads([l for l in open('gpurun_out/round6_bench.json').read().splitlines() if l.startswith('{')][-1])
r = j['roofline']
out = {k: dict(r['all_kernels'][k], **r.get('gn_traffic', {}).get(k, {})) for k in ('group_norm_fwd (HBM)', 'group_norm_bwd (HBM)') if k in r['all_kernels']}
out['note'] = ('measured = 2 * FETCH_SIZE + WRITE_SIZE of the GroupNorm kernel families (rocprofv3 --pmc, separate passes, eager steps of the bench command); '
               'algorithmic = the bytes the bench line prices the passes at; HBM: 8 TB/s peak, ~6.3 TB/s for a streaming copy')
json.dump(out, open('gpurun_out/round6_gn_traffic.json', 'w'), indent=1)
PY
# the parity-grade mode: bench line + kernel-trace stats of eager steps + PMC of its two kernels on the largest layer
python bench.py --dtype bf16x3 --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --traffic off --sustain-s 0 --no-calibration 2>/dev/null | tail -1 > gpurun_out/round6_bench_bf16x3.json
tools/cfg_kt.sh round6_bf16x3 --dtype bf16x3 --batch 32 --no-calibration
tools/pmc_x3.sh "128 128 256 3 0" > gpurun_out/round6_pmc_x3_128to128_256sq.txt 2>&1
python tools/thin_bench.py 32 > gpurun_out/round6_thin_f32_bench.txt 2>&1
# config 4 again (kernel stats of eager steps + graph times)
tools/gan_kt.sh; cp gpurun_out/gan_kernel_stats.csv gpurun_out/round6_config4_kernel_stats.csv
python tools/gan_graph_times.py > gpurun_out/round6_config4_graph_times.txt 2>/dev/null
