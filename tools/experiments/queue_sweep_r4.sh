#!/bin/bash
# Run on the GPU box: work-queue granularity of the headline launch with the round-4 kernels (environment knobs, no rebuild).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4v/queue.txt; mkdir -p $(dirname $OUT); : > $OUT
for c in 2 3 4 5 8; do
  for t in 0 4 8; do
    v=$(ARB_QUEUE_CHUNK=$c ARB_QUEUE_TAIL=$t ARB_BENCH_LEGS=perstep python3 $R/bench.py --steps 40 --warmup 40 --min-seconds 1.5 --no-cpu-baseline --no-per-step-leg 2>/dev/null | python3 -c "import sys,json; print('%.2f' % (json.loads(sys.stdin.readline())['value']/1e6))") || exit 1
    echo "chunk $c tail $t: $v M" | tee -a $OUT
  done
done
