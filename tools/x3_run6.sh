cd /root/repo
python bench.py --steps 20 --warmup 5 --quick --no-other-configs > gpurun_out/b6.json 2> gpurun_out/b6.err
tail -3 gpurun_out/b6.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/b6.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], j['value'], j['step_roofline'])
r=j['roofline']
print({k:r[k] for k in ('achieved','frac','traffic','avg_kernel_launch_us')})
for k in ('group_norm_fwd (HBM)','group_norm_bwd (HBM)'):
    print(k, r['all_kernels'][k])
print(r.get('gn_traffic'))
print(j['cpu_baseline'])
print(j['bf16_vs_fp32_oracle'])
PY
