#!/bin/bash
# Run on the GPU box: like tools/ab_multi4.sh with an environment assignment for every run and the one-step leg printed.
# usage: tools/ab_multi_env.sh <out-name> <VAR=value> <lib> [<lib> ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1.txt; E=$2; shift 2; mkdir -p $(dirname $OUT); : > $OUT
for round in 1 2; do
  for lib in "$@"; do
    v=$(env $E ARB_BENCH_LEGS=perstep ARBSTEP_LIB=$R/build/ab/$lib.so python3 $R/bench.py --steps 40 --warmup 40 --min-seconds 1.5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f M, one-step launches %.3f M' % (d['value']/1e6, d['per_step_launch']['value']/1e6))") || exit 1
    echo "round $round [$E] $lib: $v" | tee -a $OUT
  done
done
