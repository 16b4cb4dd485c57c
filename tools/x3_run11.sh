cd /root/repo
export VQK_BENCH_ONE_GPU=1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 --batch 16 --no-calibration > gpurun_out/dry2.json 2> gpurun_out/dry2.err
echo rc=$?
tail -5 gpurun_out/dry2.err
python - <<'PY'
import json
try:
    j=json.loads([l for l in open('gpurun_out/dry2.json').read().splitlines() if l.startswith('{')][-1])
    print(j['value'], j['ms_per_step'], j['n_gpus'], j.get('degraded'))
    print(j['comm'])
    print(j['config']['launch'])
except Exception as e: print('ERR', e)
PY
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 6 --warmup 3 --batch 8 --gan --no-calibration > gpurun_out/dry2gan.json 2> gpurun_out/dry2gan.err
echo rc=$?
tail -3 gpurun_out/dry2gan.err
python - <<'PY'
import json
try:
    j=json.loads([l for l in open('gpurun_out/dry2gan.json').read().splitlines() if l.startswith('{')][-1])
    print(j['value'], j['ms_per_step'], j['n_gpus'], j.get('degraded'))
    print(j['comm'])
except Exception as e: print('ERR', e)
PY
