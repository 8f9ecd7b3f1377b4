#!/bin/bash
# Run on the GPU box: A/B of development builds under both wave pins (ARB_FORCE_WAVES=2|3), interleaved, 2 rounds.
# usage: tools/ab_bench2.sh <out-name> <lib-a> <lib-b> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1.txt; A=$2; B=$3; shift 3
: > $OUT
for round in 1 2; do
  for wv in 2 3; do
  for lib in $A $B; do
    v=$(ARB_FORCE_WAVES=$wv ARBSTEP_LIB=$R/build/ab/$lib.so python3 $R/bench.py --steps 40 --warmup 40 --min-seconds 2 --no-cpu-baseline --no-per-step-leg "$@" 2>/dev/null | python3 -c "import sys,json; print('%.3f' % (json.loads(sys.stdin.readline())['value']/1e6))") || exit 1
    echo "round $round waves $wv $lib: $v M" | tee -a $OUT
  done; done
done
