#!/bin/bash
# Run on the GPU box: A/B of development builds in the LATENCY regimes -- one launch per step (4096 worlds) and the MPC shape
# (2048 rollouts x 32 steps, one GPU).  usage: tools/ab_lat.sh <out-name> <lib> [<lib> ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1.txt; shift; mkdir -p $(dirname $OUT); : > $OUT
for round in 1 2; do
  for lib in "$@"; do
    a=$(ARBSTEP_LIB=$R/build/ab/$lib.so python3 $R/bench.py --config 5 --mpc --batch 2048 --steps 32 --warmup 32 --min-seconds 1 --no-cpu-baseline --no-per-step-leg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('mpc2048 %.3f M (%.3f ms/horizon)' % (d['value']/1e6, d['roofline']['kernel_ms']))") || exit 1
    b=$(ARBSTEP_LIB=$R/build/ab/$lib.so ARB_BENCH_LEGS=perstep python3 $R/bench.py --steps 40 --warmup 40 --min-seconds 0.5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('per-step %.3f M' % (d['per_step_launch']['value']/1e6))") || exit 1
    echo "round $round $lib: $a; $b" | tee -a $OUT
  done
done
