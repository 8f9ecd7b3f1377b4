#!/bin/bash
# GPU box: work-queue granularity at 4096 worlds after the round-4 kernel changes (ARB_QUEUE_CHUNK x ARB_QUEUE_TAIL).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/queue; mkdir -p $O; L=$O/sweep.txt; : > $L
B="python3 $R/bench.py --steps 40 --warmup 40 --min-seconds 1.5 --no-cpu-baseline --no-per-step-leg"
v() { python3 -c "import sys,json; print('%.3f' % (json.loads(sys.stdin.readline())['value']/1e6))"; }
for c in 2 4 5 8 10; do for t in 0 2 4 8; do
  echo "chunk $c tail $t: $(ARB_QUEUE_CHUNK=$c ARB_QUEUE_TAIL=$t $B 2>/dev/null | v) M" >> $L || exit 1
done; done
cat $L
