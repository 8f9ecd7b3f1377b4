#!/bin/bash
# Run on the GPU box: development builds (tools/quick_build.sh) on the headline workload, interleaved, 2 rounds.
# usage: tools/ab_multi4.sh <out-name> <lib> [<lib> ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1.txt; shift; mkdir -p $(dirname $OUT); : > $OUT
for round in 1 2; do
  for lib in "$@"; do
    v=$(ARBSTEP_LIB=$R/build/ab/$lib.so python3 $R/bench.py --steps 40 --warmup 40 --min-seconds 2 --no-cpu-baseline --no-per-step-leg 2>/dev/null | python3 -c "import sys,json; print('%.3f' % (json.loads(sys.stdin.readline())['value']/1e6))") || exit 1
    echo "round $round $lib: $v M" | tee -a $OUT
  done
done
