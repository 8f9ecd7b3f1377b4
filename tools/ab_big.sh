#!/bin/bash
# Run on the GPU box: A/B of development builds at 4096 worlds (the headline) AND at 65 536 worlds (the throughput regime).
# usage: tools/ab_big.sh <out-name> <lib> [<lib> ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1.txt; shift; mkdir -p $(dirname $OUT); : > $OUT
for round in 1 2; do
  for lib in "$@"; do
    a=$(ARBSTEP_LIB=$R/build/ab/$lib.so python3 $R/bench.py --steps 40 --warmup 40 --min-seconds 2 --no-cpu-baseline --no-per-step-leg 2>/dev/null | python3 -c "import sys,json; print('%.3f' % (json.loads(sys.stdin.readline())['value']/1e6))") || exit 1
    b=$(ARBSTEP_LIB=$R/build/ab/$lib.so python3 $R/bench.py --batch 65536 --steps 40 --warmup 40 --min-seconds 2 --no-cpu-baseline --no-per-step-leg 2>/dev/null | python3 -c "import sys,json; print('%.3f' % (json.loads(sys.stdin.readline())['value']/1e6))") || exit 1
    echo "round $round $lib: 4096 worlds $a M, 65536 worlds $b M" | tee -a $OUT
  done
done
