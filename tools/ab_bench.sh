#!/bin/bash
# A/B of development builds on the GPU box: final-state comparison + bench.py for each library given.
# usage (through gpurun): bash tools/ab_bench.sh build/libq_old.so build/libq_new.so
R=${GRAFT_REPO_ROOT:-/root/repo}
first=""
for lib in "$@"; do
  export ARBSTEP_LIB=$R/$lib
  if [ -z "$first" ]; then timeout -k 10 120 python tools/pack_check.py /tmp/ab_ref.npz || exit 1; first=1
  else timeout -k 10 120 python tools/pack_check.py /tmp/ab_new.npz /tmp/ab_ref.npz | grep -v "snap [12]" || exit 1; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'M world-steps/s %.3f' % (d['value']/1e6), 'episode launch ms %.3f' % d['roofline']['kernel_ms'], 'per-step launches %.3f M, %.4f ms' % (d['per_step_launch']['value']/1e6, d['per_step_launch']['kernel_ms']))" || exit 1
done
