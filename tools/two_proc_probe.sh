#!/bin/bash
# upper-bound probe for micro-batch pipelining: two processes, each the headline step at HALF the batch, on one GPU at the same
# time (their queues run concurrently) against one process at the full batch
R=${GRAFT_REPO_ROOT:-/root/repo}
A="--no-cpu-baseline --no-kernel-events --no-other-configs --traffic off --sustain-s 0 --warmup 10"
python $R/bench.py --batch 32 --steps 100 $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('one process  bs32:', d['ms_per_step'], 'ms', d['value'], 'img/s')"
python $R/bench.py --batch 16 --steps 100 $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('one process  bs16:', d['ms_per_step'], 'ms', d['value'], 'img/s')"
python $R/bench.py --batch 16 --steps 300 $A > /tmp/p1.log 2>/dev/null &
P1=$!
python $R/bench.py --batch 16 --steps 300 $A > /tmp/p2.log 2>/dev/null &
P2=$!
wait $P1 $P2
for f in /tmp/p1.log /tmp/p2.log; do tail -1 $f | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('two processes bs16 each:', d['ms_per_step'], 'ms', d['value'], 'img/s')"; done
