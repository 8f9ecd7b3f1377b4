#!/bin/bash
# Run on the GPU box: the one-launch-per-step leg of the bench (4096 worlds) with each build of the float32 kernel pinned.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-perstep}.txt; mkdir -p $(dirname $OUT); : > $OUT
for v in "" "ARB_FORCE_WAVES=3" "ARB_FORCE_SPEC=0"; do
  for rep in 1 2; do
    r=$(env $v python3 $R/bench.py --steps 40 --warmup 40 --min-seconds 0.5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f M episodes, per-step %.3f M (%.4f ms)' % (d['value']/1e6, d['per_step_launch']['value']/1e6, d['per_step_launch']['kernel_ms']))") || exit 1
    echo "[$v] $r" | tee -a $OUT
  done
done
