#!/bin/bash
# Run on the GPU box: the work queue's chunk / tail parameters against the headline workload (and config 5).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-qsweep}.txt
: > $OUT
for cfg in 3 5; do
for chunk in 2 3 4 5 6 8; do
for tail in 2 4 8; do
  v=$(ARB_QUEUE_CHUNK=$chunk ARB_QUEUE_TAIL=$tail python3 $R/bench.py --config $cfg --steps 40 --warmup 40 --min-seconds 1.5 --no-cpu-baseline --no-per-step-leg 2>/dev/null | python3 -c "import sys,json; print('%.3f' % (json.loads(sys.stdin.readline())['value']/1e6))") || exit 1
  echo "config $cfg chunk $chunk tail $tail: $v M" | tee -a $OUT
done; done; done
