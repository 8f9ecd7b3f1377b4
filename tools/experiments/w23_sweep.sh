#!/bin/bash
# Run on the GPU box: two-wave vs three-wave build of the headline workload over batch sizes (ARB_FORCE_WAVES).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-w23}.txt; : > $OUT
for b in 2048 2560 3072 3584 4096 6144 8192; do
  line="batch $b:"
  for wv in 2 3; do
    v=$(ARB_FORCE_WAVES=$wv python3 $R/bench.py --batch $b --steps 40 --warmup 40 --min-seconds 1.5 --no-cpu-baseline --no-per-step-leg 2>/dev/null | python3 -c "import sys,json; print('%.2f' % (json.loads(sys.stdin.readline())['value']/1e6))") || exit 1
    line="$line  waves $wv: $v M"
  done
  echo "$line" | tee -a $OUT
done
