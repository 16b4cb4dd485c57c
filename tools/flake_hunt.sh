#!/bin/bash
# run the whole GPU suite N times (no -x), keep every failure's report: tools/flake_hunt.sh [N]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
N=${1:-3}
for i in $(seq 1 $N); do
  timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v "Warning\|warn(" > gpurun_out/flake_run_$i.log
  echo "run $i: $(tail -1 gpurun_out/flake_run_$i.log)"
  grep -n "^FAILED\|^ERROR" gpurun_out/flake_run_$i.log
done
