#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4v/w23.txt; : > $OUT
for b in 2048 2560 3072 3584 4096; do
  for w in 2 3; do
    v=$(ARB_FORCE_WAVES=$w ARB_BENCH_LEGS=perstep python3 $R/bench.py --batch $b --steps 40 --warmup 40 --min-seconds 1 --no-cpu-baseline --no-per-step-leg 2>/dev/null | python3 -c "import sys,json; print('%.2f' % (json.loads(sys.stdin.readline())['value']/1e6))") || exit 1
    echo "batch $b waves $w: $v M" | tee -a $OUT
  done
done
